// TEST DOUBLE of the libsnark gadgetlib1 surface that include/ethsnarks_hip/stubs.hpp touches in its "libsnark is on the
// include path" branch: protoboard<F> (constraint_system, values, val, primary_input, add_r1cs_constraint), pb_variable<F>,
// r1cs_constraint<F>, linear combinations with getTerms() / getCoeff(), constraints held by owning pointers as in the
// reference's fork (src/export.cpp:157-190).  It exists only so that tests/test_cpp_frontend.py can COMPILE AND RUN that
// branch of the adapter in a tree where libsnark is an empty submodule; it is not libsnark, implements no prover, and is
// never shipped.  The classes are the stand-alone front end of ethsnarks_hip/circuit.hpp under libsnark's names.
#pragma once
#define ethsnarks zkhip_api_double                 // circuit.hpp's namespace, renamed for this translation unit
#include "../../../../../include/ethsnarks_hip/circuit.hpp"
#undef ethsnarks
namespace libsnark {
template <class F> using protoboard = zkhip_api_double::ProtoboardT;
template <class F> using pb_variable = zkhip_api_double::VariableT;
template <class F> using pb_variable_array = zkhip_api_double::VariableArrayT;
template <class F> using pb_linear_combination = zkhip_api_double::LinearCombinationT;
template <class F> using pb_linear_combination_array = std::vector<zkhip_api_double::LinearCombinationT>;
template <class F> using linear_term = zkhip_api_double::LinearTermT;
template <class F> using r1cs_constraint = zkhip_api_double::ConstraintT;
template <class F> using gadget = zkhip_api_double::GadgetT;
}  // namespace libsnark
