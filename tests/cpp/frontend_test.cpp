// tests/cpp/frontend_test.cpp -- the C++ side of the boundary, exercised the way the reference's own gadget tests
// do it (src/test/test_mimc.cpp, test_mimc_hash.cpp, test_merkle_tree.cpp: build the gadget on a protoboard, check
// the known answer, is_satisfied(), then stub_test_proof_verify(pb)).  Compiled by tests/test_cpp_frontend.py with
// g++ against include/ethsnarks_hip/{stubs,circuit,gadgets}.hpp and libzkhip.so (no libsnark on this machine, so
// the adapter runs on the stand-alone front end).
//
//   frontend_test kat                                  known answers only (no GPU)
//   frontend_test dump  <r1cs.json> <witness.json>     depth-29 Merkle membership circuit in r1cs2json / witness2json form
//   frontend_test prove <pk.raw> <vk.json> <proof.json>   keygen + prove + verify of that circuit through the adapter (GPU)
//   frontend_test roundtrip                            stub_test_proof_verify on the MiMC hash circuit (GPU)
//   frontend_test context <pk.raw> <vk.json>           the ProverContextT caller sequence of SURVEY section 3 (GPU)
//   frontend_test pipeline <pk.raw> <vk.json>          ProverPipeline: proofs of several witnesses kept in flight, in order (GPU)
//   frontend_test main_cli <pk.raw> <vk.json> <proof.json>   stub_main_genkeys / stub_main_prove / stub_main_verify on a module class (GPU)
//   frontend_test verify_cli <vk.json> <proof.json>    stub_main_verify (host only): exit code 0 / 1 / 2 like the reference's
#include "ethsnarks_hip/stubs.hpp"
#include "ethsnarks_hip/gadgets.hpp"

#include <fstream>

using namespace ethsnarks;

static const char *ITEM_A = "3703141493535563179657531719960160174296085208671919316200479060314459804651";
static const char *ITEM_B = "134551314051432487569247388144051420116740427803855572138106146683954151557";

#define CHECK(cond, name) do { if (!(cond)) { std::cerr << "FAIL " << name << std::endl; return false; } } while (0)

static bool test_mimc() {                                   // src/test/test_mimc.cpp:45-66
    ProtoboardT pb;
    const VariableT in_x = make_variable(pb, FieldT(ITEM_A), "x");
    const VariableT in_k = make_variable(pb, FieldT(ITEM_B), "k");
    pb.set_input_sizes(2);
    MiMC_e7_gadget the_gadget(pb, in_x, in_k, "gadget");
    the_gadget.generate_r1cs_witness();
    the_gadget.generate_r1cs_constraints();
    const FieldT expected("11437467823393790387399137249441941313717686441929791910070352316474327319704");
    CHECK(pb.val(the_gadget.result()) == expected, "mimc cipher known answer");
    CHECK(pb.num_constraints() == 91 * 4, "mimc constraint count");
    CHECK(pb.is_satisfied(), "mimc is_satisfied");
    CHECK(mimc(FieldT(1), FieldT(1)) == FieldT("2447343676970420247355835473667983267115132689045447905848734383579598297563"), "mimc(1,1)");
    CHECK(mimc_hash({FieldT(1), FieldT(1)}) == FieldT("4087330248547221366577133490880315793780387749595119806283278576811074525767"), "mimc_hash([1,1])");
    return true;
}

static bool build_mimc_hash(ProtoboardT &pb) {              // src/test/test_mimc_hash.cpp:11-70
    const VariableT m_0 = make_variable(pb, FieldT(ITEM_A), "m_0");
    const VariableT m_1 = make_variable(pb, FieldT(ITEM_B), "m_1");
    pb.set_input_sizes(2);
    const VariableT iv = make_variable(pb, FieldT("918403109389145570117360101535982733651217667914747213867238065296420114726"), "iv");
    MiMC_e7_hash_gadget the_gadget(pb, iv, {m_0, m_1}, "gadget");
    the_gadget.generate_r1cs_witness();
    the_gadget.generate_r1cs_constraints();
    CHECK(pb.val(the_gadget.result()) == FieldT("15683951496311901749339509118960676303290224812129752890706581988986633412003"), "mimc hash known answer");
    CHECK(pb.num_constraints() == 2 * (91 * 4 + 1), "mimc hash constraint count");
    CHECK(pb.is_satisfied(), "mimc hash is_satisfied");
    return true;
}

static bool test_merkle_path_selector(int is_right) {       // src/test/test_merkle_tree.cpp:9-54
    ProtoboardT pb;
    const FieldT value_A("149674538925118052205057075966660054952481571156186698930522557832224430770");
    const FieldT value_B("9670701465464311903249220692483401938888498641874948577387207195814981706974");
    const VariableT var_A = make_variable(pb, value_A, "var_A"), var_B = make_variable(pb, value_B, "var_B");
    const VariableT var_is_right = make_variable(pb, "var_is_right");
    pb.val(var_is_right) = is_right ? 1 : 0;
    merkle_path_selector selector(pb, var_A, var_B, var_is_right, "test_merkle_path_selector");
    selector.generate_r1cs_witness();
    selector.generate_r1cs_constraints();
    CHECK(pb.val(selector.left()) == (is_right ? value_B : value_A), "selector left");
    CHECK(pb.val(selector.right()) == (is_right ? value_A : value_B), "selector right");
    CHECK(pb.is_satisfied(), "selector is_satisfied");
    return true;
}

static bool test_merkle_path_authenticator() {              // src/test/test_merkle_tree.cpp:57-110 (depth 1, right leaf)
    ProtoboardT pb;
    VariableArrayT address_bits; address_bits.allocate(pb, 1, "address_bits"); pb.val(address_bits[0]) = 1;
    VariableArrayT path; path.allocate(pb, 1, "path"); pb.val(path[0]) = FieldT(ITEM_A);
    VariableT leaf; leaf.allocate(pb, "leaf"); pb.val(leaf) = FieldT(ITEM_B);
    VariableT expected_root; expected_root.allocate(pb, "expected_root");
    pb.val(expected_root) = FieldT("3075442268020138823380831368198734873612490112867968717790651410945045657947");
    merkle_path_authenticator<MiMC_e7_hash_gadget> auth(pb, 1, address_bits, merkle_tree_IVs(pb), leaf, expected_root, path, "authenticator");
    auth.generate_r1cs_witness();
    auth.generate_r1cs_constraints();
    CHECK(auth.is_valid(), "authenticator is_valid");
    CHECK(pb.is_satisfied(), "authenticator is_satisfied");
    const auto ivs = merkle_tree_IV_values(29);             // src/gadgets/merkle_tree.cpp:78-108, first, second and last entry
    CHECK(ivs[0] == FieldT("149674538925118052205057075966660054952481571156186698930522557832224430770"), "IV[0]");
    CHECK(ivs[1] == FieldT("9670701465464311903249220692483401938888498641874948577387207195814981706974"), "IV[1]");
    CHECK(ivs[28] == FieldT("6037428193077828806710267464232314380014232668931818917272972397574634037180"), "IV[28]");
    return true;
}

// ethsnarks/merkletree.py `unique`: sha256(be16(depth) || be30(index)) mod r -- the placeholder siblings of the KAT tree
static FieldT merkle_unique(unsigned depth, unsigned long index) {
    uint8_t msg[32] = {0};
    msg[0] = (uint8_t)(depth >> 8); msg[1] = (uint8_t)depth;
    for (int k = 0; k < 8; k++) msg[31 - k] = (uint8_t)(index >> (8 * k));
    hashes::sha256 h; h.update(msg, 32);
    uint8_t d[32]; h.digest(d);
    return FieldT::from_bytes_be(d);
}

// BASELINE config 4: depth-29 membership proof of leaf 0 of the reference's known-answer tree (test/test_merkle.py:82-107);
// the root is the one public input.  Allocation order = ethsnarks_amd/gadgets.py merkle_membership_circuit.
static bool build_merkle29(ProtoboardT &pb) {
    const size_t depth = 29;
    VariableT expected_root; expected_root.allocate(pb, "expected_root");
    pb.set_input_sizes(1);
    VariableArrayT address_bits; address_bits.allocate(pb, depth, "address_bits"); address_bits.fill_with_bits_of_ulong(pb, 0);
    VariableArrayT path; path.allocate(pb, depth, "path");
    pb.val(path[0]) = FieldT(ITEM_B);
    for (size_t d = 1; d < depth; d++) pb.val(path[d]) = merkle_unique((unsigned)d, 1);
    VariableT leaf; leaf.allocate(pb, "leaf"); pb.val(leaf) = FieldT(ITEM_A);
    pb.val(expected_root) = FieldT("14972246236048249827985830600768475898195156734731557762844426864943654467818");   // test/test_merkle.py:92
    merkle_path_authenticator<MiMC_e7_hash_gadget> auth(pb, depth, address_bits, merkle_tree_IVs(pb), leaf, expected_root, path, "authenticator");
    auth.generate_r1cs_witness();
    auth.generate_r1cs_constraints();
    CHECK(auth.is_valid(), "depth-29 root known answer");
    CHECK(pb.num_constraints() == 21345, "depth-29 constraint count");
    CHECK(pb.is_satisfied(), "depth-29 is_satisfied");
    return true;
}

// a circuit module in the shape stub_genkeys<GadgetT> / stub_main_prove<GadgetT> expect (src/stubs.hpp:23-55): constructed on a
// protoboard with a name, generate_r1cs_constraints(), generate_r1cs_witness()
class hash_preimage_module {
 public:
    hash_preimage_module(ProtoboardT &pb, const std::string &name)
        : m_0(make_variable(pb, FieldT(ITEM_A), "m_0")), m_1(make_variable(pb, FieldT(ITEM_B), "m_1")),
          iv((pb.set_input_sizes(2), make_variable(pb, FieldT("918403109389145570117360101535982733651217667914747213867238065296420114726"), "iv"))),
          hash(pb, iv, {m_0, m_1}, name + ".hash") {}
    void generate_r1cs_constraints() { hash.generate_r1cs_constraints(); }
    void generate_r1cs_witness() { hash.generate_r1cs_witness(); }
 private:
    const VariableT m_0, m_1, iv;
    MiMC_e7_hash_gadget hash;
};

static bool write_text(const char *path, const std::string &s) { std::ofstream f(path, std::ios::binary); f << s; return (bool)f; }

int main(int argc, char **argv) {
    ppT::init_public_params();
    const std::string mode = argc > 1 ? argv[1] : "kat";
    try {
        if (mode == "kat") {
            ProtoboardT h, m;
            const bool ok = test_mimc() && build_mimc_hash(h) && test_merkle_path_selector(0) && test_merkle_path_selector(1) &&
                            test_merkle_path_authenticator() && build_merkle29(m);
            std::cout << (ok ? "OK" : "FAIL") << std::endl;
            return ok ? 0 : 1;
        }
        if (mode == "dump" && argc == 4) {
            ProtoboardT pb;
            if (!build_merkle29(pb)) return 1;
            return write_text(argv[2], r1cs2json(pb)) && write_text(argv[3], witness2json(pb)) ? 0 : 1;
        }
        if (mode == "prove" && argc == 5) {
            ProtoboardT pb;
            if (!build_merkle29(pb)) return 1;
            if (stub_genkeys_from_pb(pb, argv[2], argv[3]) != 0) { std::cerr << "genkeys failed: " << zk_last_error() << std::endl; return 1; }
            const std::string proof = stub_prove_from_pb(pb, argv[2]);
            if (!write_text(argv[4], proof)) return 1;
            std::ifstream vf(argv[3], std::ios::binary);
            const std::string vk((std::istreambuf_iterator<char>(vf)), std::istreambuf_iterator<char>());
            const bool ok = stub_verify(vk.c_str(), proof.c_str());
            std::cout << (ok ? "VERIFIED" : "REJECTED") << std::endl;
            return ok ? 0 : 1;
        }
        if (mode == "verify_cli") {                           // stub_main_verify's own argv convention: {command, vk, proof}
            std::vector<const char *> av;
            for (int i = 1; i < argc; i++) av.push_back(argv[i]);
            return stub_main_verify("frontend_test", (int)av.size(), av.data());
        }
        if (mode == "context" && argc == 4) {                 // the caller sequence of SURVEY section 3 (P), member for member
            ProtoboardT pb;
            if (!build_mimc_hash(pb)) return 1;
            if (stub_genkeys_from_pb(pb, argv[2], argv[3]) != 0) return 1;
            ProvingKeyT pk = load_proving_key(argv[2]);
            ProverContextT ctx(pk);
            ctx.constraint_system = &pb.constraint_system;
            ctx.config = libsnark::Config();
            ctx.domain = get_domain(pb, pk, ctx.config);
            const std::string json = prove(ctx, pb);
            std::ifstream vf(argv[3], std::ios::binary);
            const std::string vk((std::istreambuf_iterator<char>(vf)), std::istreambuf_iterator<char>());
            const bool ok = ctx.domain->m == 1024 && stub_verify(vk.c_str(), json.c_str()) && prove(ctx, pb) == json;
            std::cout << (ok ? "VERIFIED" : "REJECTED") << std::endl;
            return ok ? 0 : 1;
        }
        if (mode == "pipeline" && argc == 4) {                // ProverPipeline: 7 witnesses of one circuit through 2 contexts, in order
            ProtoboardT pb;
            const VariableT m_0 = make_variable(pb, FieldT(ITEM_A), "m_0"), m_1 = make_variable(pb, FieldT(ITEM_B), "m_1");
            pb.set_input_sizes(2);
            const VariableT iv = make_variable(pb, FieldT("918403109389145570117360101535982733651217667914747213867238065296420114726"), "iv");
            MiMC_e7_hash_gadget the_gadget(pb, iv, {m_0, m_1}, "gadget");
            the_gadget.generate_r1cs_witness();
            the_gadget.generate_r1cs_constraints();
            if (stub_genkeys_from_pb(pb, argv[2], argv[3]) != 0) return 1;
            ProvingKeyT pk = load_proving_key(argv[2]);
            ProverContextT ctx(pk);
            ProverPipeline pipe(pk, pb, 2);
            std::vector<std::string> expect, got;
            bool ok = true;
            for (int round = 0; round < 2; round++) {
                const int n = round ? 3 : 4;                      // 4 = 2 running + 2 staged (full), then 3
                for (int i = 0; i < n; i++) {
                    pb.val(m_0) = FieldT(ITEM_A) + FieldT(1000 * round + i);
                    the_gadget.generate_r1cs_witness();
                    ok = ok && pb.is_satisfied();
                    expect.push_back(prove(ctx, pb));
                    pipe.submit(pb);
                }
                if (!round) {
                    ok = ok && pipe.full();
                    try { pipe.submit(pb); ok = false; } catch (const std::runtime_error &) {}
                }
                while (pipe.pending()) got.push_back(pipe.next());
            }
            std::ifstream vf(argv[3], std::ios::binary);
            const std::string vk((std::istreambuf_iterator<char>(vf)), std::istreambuf_iterator<char>());
            ok = ok && got == expect && got.size() == 7 && got[0] != got[1] && stub_verify(vk.c_str(), got[6].c_str());
            std::cout << (ok ? "VERIFIED" : "REJECTED") << std::endl;
            return ok ? 0 : 1;
        }
        if (mode == "main_cli" && argc == 5) {                // stub_main_genkeys / stub_main_prove / stub_main_verify, the reference's CLI helpers
            char cmd_g[] = "genkeys", cmd_p[] = "prove";
            char *g_args[] = {cmd_g, argv[2], argv[3]}, *p_args[] = {cmd_p, argv[2], argv[4]};
            const char *v_args[] = {"verify", argv[3], argv[4]};
            bool ok = stub_main_genkeys<hash_preimage_module>("frontend_test", 2, g_args) == 1;      // usage error
            ok = ok && stub_main_prove<hash_preimage_module>("frontend_test", 2, p_args) == 1;
            ok = ok && stub_main_genkeys<hash_preimage_module>("frontend_test", 3, g_args) == 0;
            ok = ok && stub_main_prove<hash_preimage_module>("frontend_test", 3, p_args) == 0;
            ok = ok && stub_main_verify("frontend_test", 3, v_args) == 0;
            std::cout << (ok ? "VERIFIED" : "REJECTED") << std::endl;
            return ok ? 0 : 1;
        }
        if (mode == "roundtrip") {
            ProtoboardT pb;
            if (!build_mimc_hash(pb)) return 1;
            const ProtoboardT &cpb = pb;                      // stub_test_proof_verify takes a const protoboard (src/stubs.hpp:14)
            const bool ok = stub_test_proof_verify(cpb);
            std::cout << (ok ? "VERIFIED" : "REJECTED") << std::endl;
            return ok ? 0 : 1;
        }
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 2;
    }
    std::cerr << "usage: frontend_test kat | dump <r1cs.json> <witness.json> | prove <pk.raw> <vk.json> <proof.json> | roundtrip" << std::endl;
    return 1;
}
