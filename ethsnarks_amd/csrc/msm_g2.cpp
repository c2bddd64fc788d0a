// msm_g2.cpp -- G2 instantiation of the MSM kernels (B-query; tcc:499-506)
#include "msm_impl.hpp"
template struct zk::MsmWork<zk::G2>;
