// stands for the reference's src/prover_config.hpp included AFTER the adapter: same include guard, so its struct Config must
// be skipped (a second definition here would be a redefinition error)
#ifndef ETHSNARKS_PROVER_CONFIG_HPP_
#define ETHSNARKS_PROVER_CONFIG_HPP_
namespace libsnark { struct Config { int this_definition_must_not_be_seen; }; }
#endif
