// msm_g2.cpp -- G2 instantiations: MSM kernels (B-query; tcc:499-506) and the fixed-base batch
// multiplication of the key generator (tcc:394)
#include "msm_impl.hpp"
#include "keygen_impl.hpp"
template struct zk::MsmWork<zk::G2>;
template int zk::batch_mul_base<zk::G2>(const zk::G2::Affine &, const zk::fe *, uint32_t, zk::G2::Affine *, hipStream_t);
