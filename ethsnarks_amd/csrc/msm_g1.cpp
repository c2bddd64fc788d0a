// msm_g1.cpp -- G1 instantiations: MSM kernels (A-, H-, L-query; tcc:488-495,510-530) and the
// fixed-base batch multiplication of the key generator (tcc:358-411)
#include "msm_impl.hpp"
#include "keygen_impl.hpp"
template struct zk::MsmWork<zk::G1>;
template int zk::batch_mul_base<zk::G1>(const zk::G1::Affine &, const zk::fe *, uint32_t, zk::G1::Affine *, hipStream_t);
