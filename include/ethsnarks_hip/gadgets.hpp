// ethsnarks_hip/gadgets.hpp -- the two self-contained ethsnarks circuits the measurement configs name
// (BASELINE configs 1 and 4), written against the stand-alone front end of circuit.hpp:
//
//   MiMCe7_round, MiMC_gadget<RoundT>, MiMC_e7_gadget        src/gadgets/mimc.hpp:115-318
//   MiyaguchiPreneel_OWF<CipherT>, MiMC_e7_hash_gadget       src/gadgets/onewayfunction.hpp:67-127, mimc.hpp:321-346
//   merkle_path_selector                                     src/gadgets/merkle_tree.hpp:35-63, merkle_tree.cpp:11-72
//   markle_path_compute<HashT>, merkle_path_authenticator    src/gadgets/merkle_tree.hpp:71-191  (the reference spells "markle")
//   merkle_tree_IVs                                          src/gadgets/merkle_tree.cpp:75-113 (= ethsnarks/merkletree.py:36-44)
//   mimc(), mimc_hash()                                      src/gadgets/mimc.hpp:352-393 (native evaluation through a protoboard)
//
// Same class names, constructor arguments, variable allocation order and constraint order as the reference, so a
// circuit built here has the reference's shape (depth-29 authenticator: 21 345 constraints) and the reference's
// known answers hold (tests/cpp/frontend_test.cpp).  The hash primitives the constants need -- Keccak-256 (original
// 0x01 padding, the reference's SHA3IUF dependency) and SHA-256 (merkletree.py's IV chain) -- are included so
// nothing outside this header is required.  Not compiled when libsnark is present: the reference's own gadget
// headers are the ones to use then.
#pragma once
#include "circuit.hpp"

#include <cassert>
#include <mutex>

namespace ethsnarks {

// ------------------------------------------------------------------------------------------ Keccak-256 / SHA-256
namespace hashes {
inline uint64_t rotl64(uint64_t v, unsigned n) { n &= 63; return n ? (v << n) | (v >> (64 - n)) : v; }
inline void keccak_f1600(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL, 0x0000000080000001ULL,
        0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
        0x000000000000800AULL, 0x800000008000000AULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const unsigned ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};   // [x][y]
    for (int round = 0; round < 24; round++) {
        uint64_t C[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; x++) { const uint64_t D = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1); for (int y = 0; y < 5; y++) s[x + 5 * y] ^= D; }
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(s[x + 5 * y], ROT[x][y]);
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) s[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[round];
    }
}
inline void keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    std::vector<uint8_t> p(data, data + len);
    p.push_back(0x01);
    while (p.size() % rate) p.push_back(0);
    p.back() |= 0x80;
    uint64_t s[25] = {0};
    for (size_t off = 0; off < p.size(); off += rate) {
        for (size_t i = 0; i < rate / 8; i++) { uint64_t v = 0; for (int k = 7; k >= 0; k--) v = (v << 8) | p[off + 8 * i + (size_t)k]; s[i] ^= v; }
        keccak_f1600(s);
    }
    for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(s[i] >> (8 * k));
}

struct sha256 {                                              // streaming, so a running digest can be read after every update
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    std::vector<uint8_t> buf;
    uint64_t total = 0;
    static uint32_t rotr(uint32_t v, unsigned n) { return (v >> n) | (v << (32 - n)); }
    static void block(uint32_t h[8], const uint8_t *b) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
            0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
            0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
            0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
            0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
            0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t *d, size_t n) {
        total += n;
        buf.insert(buf.end(), d, d + n);
        size_t off = 0;
        for (; off + 64 <= buf.size(); off += 64) block(h, buf.data() + off);
        buf.erase(buf.begin(), buf.begin() + (long)off);
    }
    void digest(uint8_t out[32]) const {                     // non-destructive: the stream can continue afterwards
        uint32_t hc[8]; memcpy(hc, h, sizeof(hc));
        std::vector<uint8_t> p(buf);
        p.push_back(0x80);
        while (p.size() % 64 != 56) p.push_back(0);
        const uint64_t bits = total * 8;
        for (int k = 7; k >= 0; k--) p.push_back((uint8_t)(bits >> (8 * k)));
        for (size_t off = 0; off < p.size(); off += 64) block(hc, p.data() + off);
        for (int i = 0; i < 8; i++) for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(hc[i] >> (24 - 8 * k));
    }
};
}  // namespace hashes

// ------------------------------------------------------------------------------------------ MiMC
#define MIMC_SEED "mimc"

// one round: t = x + k + C;  a = t^2, b = t^4, c = t^6, d = t^7 (+ k in the last round)
class MiMCe7_round : public GadgetT {
public:
    static constexpr size_t N_ROUNDS = 91;
    const VariableT x, k;
    const FieldT C;
    const bool add_k_to_result;
    const VariableT a, b, c, d;

    MiMCe7_round(ProtoboardT &in_pb, const VariableT in_x, const VariableT in_k, const FieldT &in_C, const bool in_add_k_to_result,
                 const std::string &annotation_prefix)
        : GadgetT(in_pb, annotation_prefix), x(in_x), k(in_k), C(in_C), add_k_to_result(in_add_k_to_result),
          a(make_variable(in_pb, FMT(annotation_prefix, ".a"))), b(make_variable(in_pb, FMT(annotation_prefix, ".b"))),
          c(make_variable(in_pb, FMT(annotation_prefix, ".c"))), d(make_variable(in_pb, FMT(annotation_prefix, ".d"))) {}

    const VariableT &result() const { return d; }

    void generate_r1cs_constraints() {
        const LinearCombinationT t = x + k + C;
        pb.add_r1cs_constraint(ConstraintT(t, t, a), ".a = t^2");
        pb.add_r1cs_constraint(ConstraintT(a, a, b), ".b = t^4");
        pb.add_r1cs_constraint(ConstraintT(a, b, c), ".c = t^6");
        if (add_k_to_result) pb.add_r1cs_constraint(ConstraintT(t, c, d - k), ".d = t^7 + k");
        else pb.add_r1cs_constraint(ConstraintT(t, c, d), ".d = t^7");
    }
    void generate_r1cs_witness() const {
        const FieldT vk = pb.val(k), t = pb.val(x) + vk + C;
        const FieldT va = t * t, vb = va * va, vc = va * vb;
        pb.val(a) = va; pb.val(b) = vb; pb.val(c) = vc;
        pb.val(d) = vc * t + (add_k_to_result ? vk : FieldT::zero());
    }
};

template <typename RoundT>
class MiMC_gadget : public GadgetT {
public:
    std::vector<RoundT> m_rounds;
    const VariableT k;

    MiMC_gadget(ProtoboardT &in_pb, const VariableT in_x, const VariableT in_k, const std::vector<FieldT> &in_round_constants, const std::string &annotation_prefix)
        : GadgetT(in_pb, annotation_prefix), k(in_k) { setup(in_x, in_k, in_round_constants); }
    MiMC_gadget(ProtoboardT &in_pb, const VariableT in_x, const VariableT in_k, const std::string &annotation_prefix)
        : GadgetT(in_pb, annotation_prefix), k(in_k) { setup(in_x, in_k, static_constants()); }

    const VariableT &result() const { return m_rounds.back().result(); }
    void generate_r1cs_constraints() { for (auto &r : m_rounds) r.generate_r1cs_constraints(); }
    void generate_r1cs_witness() const { for (auto &r : m_rounds) r.generate_r1cs_witness(); }

    // C_0 = keccak256(keccak256(seed)), C_{i+1} = keccak256(C_i), read big-endian and reduced into the field
    static void constants_fill(std::vector<FieldT> &round_constants, const char *seed = MIMC_SEED) {
        uint8_t digest[32];
        hashes::keccak256(reinterpret_cast<const uint8_t *>(seed), strlen(seed), digest);
        round_constants.reserve(RoundT::N_ROUNDS);
        for (size_t i = 0; i < RoundT::N_ROUNDS; i++) {
            uint8_t next[32];
            hashes::keccak256(digest, 32, next);
            memcpy(digest, next, 32);
            round_constants.push_back(FieldT::from_bytes_be(digest));
        }
    }
    static const std::vector<FieldT> constants(const char *seed = MIMC_SEED) { std::vector<FieldT> c; constants_fill(c, seed); return c; }
    static const std::vector<FieldT> &static_constants() {
        static std::vector<FieldT> cached;
        static std::once_flag once;
        std::call_once(once, []() { constants_fill(cached); });
        return cached;
    }

private:
    void setup(const VariableT in_x, const VariableT in_k, const std::vector<FieldT> &rc) {
        m_rounds.reserve(rc.size());
        for (size_t i = 0; i < rc.size(); i++) {
            const VariableT round_x = i == 0 ? in_x : m_rounds.back().result();
            m_rounds.emplace_back(pb, round_x, in_k, rc[i], i + 1 == rc.size(), FMT(annotation_prefix, ".round[%zu]", i));
        }
    }
};
using MiMC_e7_gadget = MiMC_gadget<MiMCe7_round>;

// Miyaguchi-Preneel one-way function: k_{i+1} = k_i + E_{k_i}(m_i) + m_i
template <class CipherT>
class MiyaguchiPreneel_OWF : public GadgetT {
public:
    std::vector<CipherT> m_ciphers;
    const std::vector<VariableT> m_messages;
    const VariableArrayT m_outputs;
    const VariableT m_IV;

    MiyaguchiPreneel_OWF(ProtoboardT &in_pb, const VariableT in_IV, const std::vector<VariableT> &in_messages, const std::string &in_annotation_prefix)
        : GadgetT(in_pb, in_annotation_prefix), m_messages(in_messages),
          m_outputs(make_var_array(in_pb, in_messages.size(), FMT(in_annotation_prefix, ".outputs"))), m_IV(in_IV) {
        m_ciphers.reserve(in_messages.size());
        for (size_t i = 0; i < in_messages.size(); i++)
            m_ciphers.emplace_back(in_pb, in_messages[i], i == 0 ? in_IV : m_outputs[i - 1], FMT(in_annotation_prefix, ".cipher[%zu]", i));
    }
    const VariableT &result() const { return m_outputs[m_outputs.size() - 1]; }
    void generate_r1cs_constraints() {
        for (size_t i = 0; i < m_ciphers.size(); i++) {
            m_ciphers[i].generate_r1cs_constraints();
            const VariableT round_key = i == 0 ? m_IV : m_outputs[i - 1];
            pb.add_r1cs_constraint(ConstraintT(round_key + m_ciphers[i].result() + m_messages[i], 1, m_outputs[i]), ".out = k + E_k(m_i) + m_i");
        }
    }
    void generate_r1cs_witness() const {
        for (size_t i = 0; i < m_ciphers.size(); i++) {
            m_ciphers[i].generate_r1cs_witness();
            const FieldT round_key = i == 0 ? pb.val(m_IV) : pb.val(m_outputs[i - 1]);
            pb.val(m_outputs[i]) = round_key + pb.val(m_ciphers[i].result()) + pb.val(m_messages[i]);
        }
    }
};
template <typename G> using MiMC_hash_MiyaguchiPreneel_gadget = MiyaguchiPreneel_OWF<G>;
using MiMC_e7_hash_gadget = MiMC_hash_MiyaguchiPreneel_gadget<MiMC_e7_gadget>;

// native evaluation, done the way the reference does it: through a scratch protoboard
inline const FieldT mimc(const std::vector<FieldT> &round_constants, const FieldT &x, const FieldT &k) {
    ProtoboardT pb;
    const VariableT vx = make_variable(pb, x, "x"), vk = make_variable(pb, k, "k");
    MiMC_e7_gadget g(pb, vx, vk, round_constants, "mimc");
    g.generate_r1cs_witness();
    return pb.val(g.result());
}
inline const FieldT mimc(const FieldT &x, const FieldT &k) { return mimc(MiMC_e7_gadget::static_constants(), x, k); }
inline const FieldT mimc_hash(const std::vector<FieldT> &m, const FieldT &iv = FieldT::zero()) {
    ProtoboardT pb;
    const VariableT viv = make_variable(pb, iv, "iv");
    std::vector<VariableT> vars;
    for (const auto &v : m) vars.push_back(make_variable(pb, v, "m"));
    MiMC_e7_hash_gadget g(pb, viv, vars, "mimc_hash");
    g.generate_r1cs_witness();
    return pb.val(g.result());
}

// ------------------------------------------------------------------------------------------ Merkle path
// left = (1 - is_right) * input + is_right * pathvar;  right = is_right * input + (1 - is_right) * pathvar
class merkle_path_selector : public GadgetT {
public:
    const VariableT m_input, m_pathvar, m_is_right;
    VariableT m_left_a, m_left_b, m_left, m_right_a, m_right_b, m_right;

    merkle_path_selector(ProtoboardT &in_pb, const VariableT &in_input, const VariableT &in_pathvar, const VariableT &in_is_right, const std::string &in_annotation_prefix)
        : GadgetT(in_pb, in_annotation_prefix), m_input(in_input), m_pathvar(in_pathvar), m_is_right(in_is_right) {
        m_left_a.allocate(in_pb, FMT(annotation_prefix, ".left_a"));
        m_left_b.allocate(in_pb, FMT(annotation_prefix, ".left_b"));
        m_left.allocate(in_pb, FMT(annotation_prefix, ".left"));
        m_right_a.allocate(in_pb, FMT(annotation_prefix, ".right_a"));
        m_right_b.allocate(in_pb, FMT(annotation_prefix, ".right_b"));
        m_right.allocate(in_pb, FMT(annotation_prefix, ".right"));
    }
    void generate_r1cs_constraints() {
        pb.add_r1cs_constraint(ConstraintT(1 - m_is_right, m_input, m_left_a), "(1-is_right) * input = left_a");
        pb.add_r1cs_constraint(ConstraintT(m_is_right, m_pathvar, m_left_b), "is_right * pathvar = left_b");
        pb.add_r1cs_constraint(ConstraintT(m_left_a + m_left_b, 1, m_left), "left_a + left_b = left");
        pb.add_r1cs_constraint(ConstraintT(m_is_right, m_input, m_right_a), "is_right * input = right_a");
        pb.add_r1cs_constraint(ConstraintT(1 - m_is_right, m_pathvar, m_right_b), "(1-is_right) * pathvar = right_b");
        pb.add_r1cs_constraint(ConstraintT(m_right_a + m_right_b, 1, m_right), "right_a + right_b = right");
    }
    void generate_r1cs_witness() const {
        const FieldT r = pb.val(m_is_right), nr = FieldT::one() - r;
        pb.val(m_left_a) = nr * pb.val(m_input);
        pb.val(m_left_b) = r * pb.val(m_pathvar);
        pb.val(m_left) = pb.val(m_left_a) + pb.val(m_left_b);
        pb.val(m_right_a) = r * pb.val(m_input);
        pb.val(m_right_b) = nr * pb.val(m_pathvar);
        pb.val(m_right) = pb.val(m_right_a) + pb.val(m_right_b);
    }
    const VariableT &left() const { return m_left; }
    const VariableT &right() const { return m_right; }
};

// per-level IVs: IV_i = sha256("MerkleTree-" || le16(0) || ... || "MerkleTree-" || le16(i)) mod r (running digest)
inline std::vector<FieldT> merkle_tree_IV_values(size_t depth = 29) {
    std::vector<FieldT> out;
    hashes::sha256 h;
    for (size_t i = 0; i < depth; i++) {
        uint8_t msg[13] = {'M', 'e', 'r', 'k', 'l', 'e', 'T', 'r', 'e', 'e', '-', (uint8_t)(i & 0xff), (uint8_t)(i >> 8)};
        h.update(msg, sizeof(msg));
        uint8_t d[32];
        h.digest(d);
        out.push_back(FieldT::from_bytes_be(d));
    }
    return out;
}
inline const VariableArrayT merkle_tree_IVs(ProtoboardT &in_pb) {
    const VariableArrayT x = make_var_array(in_pb, 29, "IVs");
    x.fill_with_field_elements(in_pb, merkle_tree_IV_values(29));
    return x;
}

template <typename HashT>
class markle_path_compute : public GadgetT {
public:
    const size_t m_depth;
    const VariableArrayT m_address_bits;
    const VariableT m_leaf;
    const VariableArrayT m_path;
    std::vector<merkle_path_selector> m_selectors;
    std::vector<HashT> m_hashers;

    markle_path_compute(ProtoboardT &in_pb, const size_t in_depth, const VariableArrayT &in_address_bits, const VariableArrayT &in_IVs,
                        const VariableT in_leaf, const VariableArrayT &in_path, const std::string &in_annotation_prefix)
        : GadgetT(in_pb, in_annotation_prefix), m_depth(in_depth), m_address_bits(in_address_bits), m_leaf(in_leaf), m_path(in_path) {
        if (in_depth == 0 || in_address_bits.size() != in_depth || in_IVs.size() < in_depth || in_path.size() < in_depth)
            throw std::invalid_argument("markle_path_compute: depth, address bits, IVs and path do not agree");
        m_selectors.reserve(in_depth); m_hashers.reserve(in_depth);
        for (size_t i = 0; i < m_depth; i++) {
            const VariableT below = i == 0 ? in_leaf : m_hashers[i - 1].result();
            m_selectors.emplace_back(in_pb, below, in_path[i], in_address_bits[i], FMT(annotation_prefix, ".selector[%zu]", i));
            m_hashers.emplace_back(in_pb, in_IVs[i], std::vector<VariableT>{m_selectors[i].left(), m_selectors[i].right()}, FMT(annotation_prefix, ".hasher[%zu]", i));
        }
    }
    const VariableT result() const { return m_hashers.back().result(); }
    void generate_r1cs_constraints() {
        for (size_t i = 0; i < m_hashers.size(); i++) { m_selectors[i].generate_r1cs_constraints(); m_hashers[i].generate_r1cs_constraints(); }
    }
    void generate_r1cs_witness() const {
        for (size_t i = 0; i < m_hashers.size(); i++) { m_selectors[i].generate_r1cs_witness(); m_hashers[i].generate_r1cs_witness(); }
    }
};

template <typename HashT>
class merkle_path_authenticator : public markle_path_compute<HashT> {
public:
    const VariableT m_expected_root;
    merkle_path_authenticator(ProtoboardT &in_pb, const size_t in_depth, const VariableArrayT in_address_bits, const VariableArrayT in_IVs,
                              const VariableT in_leaf, const VariableT in_expected_root, const VariableArrayT in_path, const std::string &in_annotation_prefix)
        : markle_path_compute<HashT>(in_pb, in_depth, in_address_bits, in_IVs, in_leaf, in_path, in_annotation_prefix), m_expected_root(in_expected_root) {}
    bool is_valid() const { return this->pb.val(this->result()) == this->pb.val(m_expected_root); }
    void generate_r1cs_constraints() {
        markle_path_compute<HashT>::generate_r1cs_constraints();
        this->pb.add_r1cs_constraint(ConstraintT(this->result(), 1, m_expected_root), FMT(this->annotation_prefix, ".expected_root authenticator"));
    }
};

}  // namespace ethsnarks
