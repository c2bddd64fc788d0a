// msm_g1.cpp -- G1 instantiation of the MSM kernels (A-, H-, L-query; tcc:488-495,510-530)
#include "msm_impl.hpp"
template struct zk::MsmWork<zk::G1>;
