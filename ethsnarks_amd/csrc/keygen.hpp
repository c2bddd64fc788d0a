// keygen.hpp -- fixed-base batch exponentiation for the key generator.
//
// Replaces libff::batch_exp / libsnark::kc_batch_exp in r1cs_gg_ppzksnark_zok_generator
// (r1cs_gg_ppzksnark_zok.tcc:358-411): out[i] = scalars[i] * G for one fixed base G, one thread per
// scalar (MSB-first double-and-add in XYZZ, then one inversion to affine).  A table-driven windowed
// variant would be faster; key generation is off the proving path, so the simple form is kept.
#pragma once
#include <vector>
#include "bn254.hpp"

// verification key object of the C ABI (zk_keygen output, zk_vk_from_json / zk_vk_to_json): the members of
// r1cs_gg_ppzksnark_zok_verification_key (r1cs_gg_ppzksnark_zok.hpp), affine Montgomery
struct zk_vk {
    zk::G1::Affine alpha_g1; zk::G2::Affine beta_g2, gamma_g2, delta_g2;
    std::vector<zk::G1::Affine> gamma_abc;
};

namespace zk {

template <class C>
int batch_mul_base(const typename C::Affine &base, const fe *d_scalars_mont, uint32_t n,
                   typename C::Affine *d_out, hipStream_t st);

}  // namespace zk
