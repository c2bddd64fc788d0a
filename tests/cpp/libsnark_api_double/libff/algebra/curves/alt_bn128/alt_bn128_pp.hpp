// TEST DOUBLE, see libsnark/gadgetlib1/protoboard.hpp: libff::alt_bn128_pp / libff::Fr<ppT> as far as the adapter uses them
#pragma once
#include "../../../../libsnark/gadgetlib1/protoboard.hpp"
namespace libff {
struct alt_bn128_pp { typedef zkhip_api_double::FieldT Fp_type; static void init_public_params() {} };
template <class pp> using Fr = typename pp::Fp_type;
}  // namespace libff
