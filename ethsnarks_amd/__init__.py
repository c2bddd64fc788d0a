"""ethsnarks_amd -- MI355X-native Groth16 proving backend behind the ethsnarks proving API.

Only the hot path of zkh2018/ethsnarks lives here: src/stubs.cpp prove() ->
src/r1cs_gg_ppzksnark_zok/r1cs_gg_ppzksnark_zok.tcc:451-550.  Compute is hand-written HIP
(ethsnarks_amd/csrc, built into libzkhip.so, C ABI in include/zkhip.h); this Python package is the
thin host-side binding used by tests/ and bench.py.
"""
__version__ = "0.4.0"
