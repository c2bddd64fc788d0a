// tests/cpp/libsnark_branch_test.cpp -- the adapter's "libsnark headers are on the include path" branch, compiled against
// the API double of tests/cpp/libsnark_api_double (the reference checkout ships libsnark as an empty submodule) and written
// exactly as SURVEY section 3 reconstructs an ethsnarks caller:
//     ppT::init_public_params(); build pb; pk = load_proving_key(path); ProverContextT ctx(pk);
//     ctx.constraint_system = &pb.constraint_system; ctx.config = Config(); ctx.domain = get_domain(pb, pk, cfg);
//     json = prove(ctx, pb);
//   libsnark_branch_test compile-only                       (no GPU: types, printer, get_cpu_ranges)
//   libsnark_branch_test prove <pk.raw> <vk.json> <proof.json>   (GPU)
#include "ethsnarks_hip/stubs.hpp"
#include "prover_config_after.hpp"          // stands for a later #include of the reference's src/prover_config.hpp: must be a no-op

#include <sstream>

using namespace ethsnarks;

static_assert(ETHSNARKS_HIP_HAVE_LIBSNARK == 1, "this test must take the libsnark branch of the adapter");
static_assert(std::is_same<ProtoboardT, libsnark::protoboard<FieldT>>::value, "ProtoboardT alias (src/ethsnarks.hpp:34)");
static_assert(std::is_same<decltype(ProverContextT::config), libsnark::Config>::value, "ProverContext::config (hpp:284)");

// x * y = z, (x + z) * 1 = u, public input z
static void build(ProtoboardT &pb) {
    VariableT z, x, y, u;
    z.allocate(pb, "z"); x.allocate(pb, "x"); y.allocate(pb, "y"); u.allocate(pb, "u");
    pb.set_input_sizes(1);
    pb.val(x) = FieldT("1234567890123456789012345678901234567890"); pb.val(y) = FieldT(77L);
    pb.val(z) = pb.val(x) * pb.val(y); pb.val(u) = pb.val(x) + pb.val(z);
    pb.add_r1cs_constraint(ConstraintT(x, y, z), "mul");
    pb.add_r1cs_constraint(ConstraintT(x + z, FieldT::one(), u), "add");
}

int main(int argc, char **argv) {
    ppT::init_public_params();
    ProtoboardT pb;
    build(pb);
    if (!pb.is_satisfied()) return 2;
    {   // Config printer and get_cpu_ranges (src/prover_config.hpp:37-85)
        libsnark::Config c; c.radixes = {4, 8}; c.multi_exp_c = 16;
        std::ostringstream os; os << c;
        if (os.str() != "num_threads: 1, smt: 0, fft: recursive, radixes: [4,8], exp_c: 16, pre_stride: 128, exp_preloc: 0, exp_lookahead: 1") { std::cerr << os.str() << std::endl; return 3; }
        const auto r = libsnark::get_cpu_ranges(2, 12, 3);
        if (r.size() != 3 || r[0] != std::make_pair(2u, 6u) || r[1] != std::make_pair(6u, 9u) || r[2] != std::make_pair(9u, 12u)) return 4;
        if (!libsnark::get_cpu_ranges(5, 5).empty() || libsnark::get_cpu_ranges(0, 7).size() != 1) return 5;
    }
    if (argc >= 2 && std::string(argv[1]) == "compile-only") { std::cout << "OK" << std::endl; return 0; }
    if (argc < 5 || std::string(argv[1]) != "prove") return 1;
    if (stub_genkeys_from_pb(pb, argv[2], argv[3]) != 0) return 6;
    // SURVEY section 3, (P)
    ProvingKeyT pk = load_proving_key(argv[2]);
    ProverContextT ctx(pk);
    ctx.constraint_system = &pb.constraint_system;
    ctx.config = libsnark::Config();
    ctx.domain = get_domain(pb, pk, ctx.config);
    ctx.scratch_exponents.clear(); ctx.aA.clear(); ctx.aB.clear(); ctx.aH.clear();      // the reference's scratch members exist
    const std::string json = prove(ctx, pb);
    const std::string again = prove(ctx, pb);                                             // context reuse
    std::ofstream(argv[4], std::ios::binary) << json;
    std::ifstream vf(argv[3], std::ios::binary);
    const std::string vk((std::istreambuf_iterator<char>(vf)), std::istreambuf_iterator<char>());
    const ProtoboardT &cpb = pb;
    if (json != again || !stub_verify(vk.c_str(), json.c_str()) || !stub_test_proof_verify(cpb)) return 7;
    std::cout << "OK" << std::endl;
    return 0;
}
