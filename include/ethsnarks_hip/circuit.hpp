// ethsnarks_hip/circuit.hpp -- a small stand-alone R1CS front end with the surface ethsnarks gadget code uses
// (SURVEY 8(f)-3a): FieldT, VariableT, VariableArrayT, ConstraintT, ProtoboardT, GadgetT, make_variable,
// make_var_array, FMT.  It exists so that circuits can be written and proven through libzkhip.so on a machine that
// has no libsnark (the reference checkout ships libsnark as an empty submodule), and so that the adapter
// ethsnarks_hip/stubs.hpp can be compiled and tested without it.  With libsnark present the adapter uses
// libsnark's own protoboard and this header is not included.
//
// What it mirrors (names and meaning, not code):
//   src/ethsnarks.hpp:31-48        FieldT, ProtoboardT, VariableT, VariableArrayT, ConstraintT, GadgetT
//   src/utils.hpp / utils.cpp      make_variable, make_var_array, FMT
//   the accessor chain of src/export.cpp:157-190: pb.constraint_system.constraints[c]->getA().getTerms()
//   -> {index, getCoeff()}; pb.values = Fr[V+1] with ONE at index 0 (r1cs_gg_ppzksnark_zok.tcc:492-493)
// FieldT has the memory image of libff::Fp_model<4> for alt_bn128 Fr: 4 x u64 little-endian limbs of
// value * 2^256 mod r, which is what zk_prove / zk_ctx_create take.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

namespace ethsnarks {

// ------------------------------------------------------------------------------------------ Fr
class FieldT {
public:
    struct bigint { uint64_t data[4]; };                   // canonical value, little-endian limbs

    FieldT() : l{0, 0, 0, 0} {}
    FieldT(long v) : l{0, 0, 0, 0} {
        const bool neg = v < 0;
        uint64_t c[4] = {neg ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v, 0, 0, 0};
        *this = from_canonical(c);
        if (neg) *this = -*this;
    }
    FieldT(int v) : FieldT((long)v) {}
    FieldT(unsigned long v) : l{0, 0, 0, 0} { uint64_t c[4] = {v, 0, 0, 0}; *this = from_canonical(c); }
    explicit FieldT(const char *decimal) : l{0, 0, 0, 0} {   // like libff: numerals below the modulus; larger ones reduce
        if (!decimal || !*decimal) throw std::invalid_argument("FieldT: empty numeral");
        FieldT acc, ten(10L);
        for (const char *p = decimal; *p; p++) {
            if (*p < '0' || *p > '9') throw std::invalid_argument("FieldT: not a decimal numeral");
            acc = acc * ten + FieldT((long)(*p - '0'));
        }
        *this = acc;
    }
    explicit FieldT(const bigint &b) : l{0, 0, 0, 0} { *this = from_canonical(b.data); }
    // 32 big-endian bytes (mpz_import order 1), reduced mod r: how the MiMC round constants enter the field
    static FieldT from_bytes_be(const uint8_t b[32]) {
        uint64_t c[4];
        for (int i = 0; i < 4; i++) { uint64_t v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | b[8 * (3 - i) + k]; c[i] = v; }
        return from_canonical(c);
    }

    static FieldT zero() { return FieldT(); }
    static FieldT one() { FieldT r; for (int i = 0; i < 4; i++) r.l[i] = ONE[i]; return r; }
    static FieldT random_element() {
        static std::mt19937_64 gen{std::random_device{}()};
        uint64_t c[4] = {gen(), gen(), gen(), gen() >> 2};
        return from_canonical(c);
    }

    bool is_zero() const { return !(l[0] | l[1] | l[2] | l[3]); }
    bool operator==(const FieldT &o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
    bool operator!=(const FieldT &o) const { return !(*this == o); }

    FieldT operator+(const FieldT &o) const {
        FieldT r; unsigned __int128 c = 0;
        for (int i = 0; i < 4; i++) { c += (unsigned __int128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
        r.reduce_once((uint64_t)c);
        return r;
    }
    FieldT operator-(const FieldT &o) const {
        FieldT r; unsigned __int128 b = 0;
        for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)l[i] - o.l[i] - (uint64_t)b; r.l[i] = (uint64_t)d; b = (d >> 64) & 1; }
        if (b) { unsigned __int128 c = 0; for (int i = 0; i < 4; i++) { c += (unsigned __int128)r.l[i] + MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
        return r;
    }
    FieldT operator-() const { return is_zero() ? *this : zero() - *this; }
    FieldT operator*(const FieldT &o) const {                // coarsely integrated operand scanning, 4 x 64-bit limbs
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; j < 4; j++) { c += (unsigned __int128)l[j] * o.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * INV;
            c = ((unsigned __int128)m * MOD[0] + t[0]) >> 64;
            for (int j = 1; j < 4; j++) { c += (unsigned __int128)m * MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        FieldT r; for (int i = 0; i < 4; i++) r.l[i] = t[i];
        r.reduce_once(t[4]);
        return r;
    }
    FieldT &operator+=(const FieldT &o) { return *this = *this + o; }
    FieldT &operator-=(const FieldT &o) { return *this = *this - o; }
    FieldT &operator*=(const FieldT &o) { return *this = *this * o; }
    FieldT squared() const { return *this * *this; }
    FieldT operator^(unsigned long e) const { FieldT r = one(), b = *this; for (; e; e >>= 1) { if (e & 1) r *= b; b *= b; } return r; }
    FieldT inverse() const {                                 // a^(r-2)
        uint64_t e[4] = {MOD[0] - 2, MOD[1], MOD[2], MOD[3]};
        FieldT r = one();
        for (int i = 255; i >= 0; i--) { r *= r; if ((e[i / 64] >> (i % 64)) & 1) r *= *this; }
        return r;
    }

    bigint as_bigint() const {                               // leave Montgomery form
        FieldT u; u.l[0] = 1;
        const FieldT c = *this * u;
        bigint b; for (int i = 0; i < 4; i++) b.data[i] = c.l[i];
        return b;
    }
    std::string to_decimal() const {
        bigint b = as_bigint();
        std::string s;
        for (;;) {
            unsigned __int128 rem = 0; bool nz = false;
            for (int i = 3; i >= 0; i--) { const unsigned __int128 cur = (rem << 64) | b.data[i]; b.data[i] = (uint64_t)(cur / 10); rem = cur % 10; nz |= b.data[i] != 0; }
            s.insert(s.begin(), (char)('0' + (int)rem));
            if (!nz) break;
        }
        return s;
    }
    void print() const { std::cerr << to_decimal() << std::endl; }
    const uint64_t *limbs() const { return l; }

private:
    uint64_t l[4];
    static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t ONE[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
    static constexpr uint64_t INV = 0xc2e1f593efffffffULL;   // -r^-1 mod 2^64
    void reduce_once(uint64_t hi) {
        bool ge = hi != 0;
        if (!ge) { ge = true; for (int i = 3; i >= 0; i--) if (l[i] != MOD[i]) { ge = l[i] > MOD[i]; break; } }
        if (ge) { unsigned __int128 b = 0; for (int i = 0; i < 4; i++) { const unsigned __int128 d = (unsigned __int128)l[i] - MOD[i] - (uint64_t)b; l[i] = (uint64_t)d; b = (d >> 64) & 1; } }
    }
    static FieldT from_canonical(const uint64_t c[4]) {      // any 256-bit value: c * R^2 / R = c * R mod r
        FieldT a, r2;
        for (int i = 0; i < 4; i++) { a.l[i] = c[i]; r2.l[i] = R2[i]; }
        return a * r2;
    }
};
static_assert(sizeof(FieldT) == 32, "FieldT must have the memory image of libff::Fp_model<4>");

// ------------------------------------------------------------------------------------------ variables, linear combinations
class ProtoboardT;
struct VariableT {
    uint32_t index = 0;                                      // 0 is the constant ONE
    VariableT() = default;
    explicit VariableT(uint32_t i) : index(i) {}
    void allocate(ProtoboardT &pb, const std::string &annotation = "");
};

struct LinearTermT {
    uint32_t index;
    FieldT coeff;
    const FieldT &getCoeff() const { return coeff; }
};

class LinearCombinationT {
public:
    LinearCombinationT() = default;
    LinearCombinationT(const VariableT &v) { terms.push_back({v.index, FieldT::one()}); }
    LinearCombinationT(const FieldT &c) { if (!c.is_zero()) terms.push_back({0, c}); }
    LinearCombinationT(int c) : LinearCombinationT(FieldT((long)c)) {}
    LinearCombinationT(long c) : LinearCombinationT(FieldT(c)) {}
    const std::vector<LinearTermT> &getTerms() const { return terms; }
    // terms stay sorted by index and merged, so the same expression always gives the same row
    LinearCombinationT &add_term(uint32_t index, const FieldT &coeff) {
        size_t i = 0;
        while (i < terms.size() && terms[i].index < index) i++;
        if (i < terms.size() && terms[i].index == index) {
            terms[i].coeff += coeff;
            if (terms[i].coeff.is_zero()) terms.erase(terms.begin() + (long)i);
        } else if (!coeff.is_zero()) terms.insert(terms.begin() + (long)i, LinearTermT{index, coeff});
        return *this;
    }
    LinearCombinationT operator+(const LinearCombinationT &o) const { LinearCombinationT r = *this; for (const auto &t : o.terms) r.add_term(t.index, t.coeff); return r; }
    LinearCombinationT operator-(const LinearCombinationT &o) const { LinearCombinationT r = *this; for (const auto &t : o.terms) r.add_term(t.index, -t.coeff); return r; }
    LinearCombinationT operator*(const FieldT &k) const { LinearCombinationT r; for (const auto &t : terms) r.add_term(t.index, t.coeff * k); return r; }
    LinearCombinationT operator-() const { return LinearCombinationT() - *this; }
private:
    std::vector<LinearTermT> terms;
};
inline LinearCombinationT operator+(const VariableT &a, const VariableT &b) { return LinearCombinationT(a) + LinearCombinationT(b); }
inline LinearCombinationT operator+(const VariableT &a, const FieldT &b) { return LinearCombinationT(a) + LinearCombinationT(b); }
inline LinearCombinationT operator+(const VariableT &a, const LinearCombinationT &b) { return LinearCombinationT(a) + b; }
inline LinearCombinationT operator-(const VariableT &a, const VariableT &b) { return LinearCombinationT(a) - LinearCombinationT(b); }
inline LinearCombinationT operator-(const VariableT &a, const FieldT &b) { return LinearCombinationT(a) - LinearCombinationT(b); }
inline LinearCombinationT operator-(const VariableT &a, const LinearCombinationT &b) { return LinearCombinationT(a) - b; }
inline LinearCombinationT operator-(int a, const VariableT &b) { return LinearCombinationT(a) - LinearCombinationT(b); }
inline LinearCombinationT operator+(int a, const VariableT &b) { return LinearCombinationT(a) + LinearCombinationT(b); }
inline LinearCombinationT operator-(int a, const LinearCombinationT &b) { return LinearCombinationT(a) - b; }
inline LinearCombinationT operator*(const FieldT &k, const VariableT &v) { return LinearCombinationT(v) * k; }
inline LinearCombinationT operator*(const VariableT &v, const FieldT &k) { return LinearCombinationT(v) * k; }
inline LinearCombinationT operator*(const FieldT &k, const LinearCombinationT &v) { return v * k; }
inline LinearCombinationT operator*(int k, const VariableT &v) { return LinearCombinationT(v) * FieldT((long)k); }

class ConstraintT {                                          // <a, x> * <b, x> = <c, x>
public:
    ConstraintT(const LinearCombinationT &a, const LinearCombinationT &b, const LinearCombinationT &c) : a_(a), b_(b), c_(c) {}
    const LinearCombinationT &getA() const { return a_; }
    const LinearCombinationT &getB() const { return b_; }
    const LinearCombinationT &getC() const { return c_; }
private:
    LinearCombinationT a_, b_, c_;
};

struct ConstraintSystemT {
    std::vector<std::unique_ptr<ConstraintT>> constraints;   // owning pointers, like the fork's "light" constraints
    size_t primary_input_size = 0, auxiliary_input_size = 0;
    size_t num_constraints() const { return constraints.size(); }
    size_t num_inputs() const { return primary_input_size; }
    size_t num_variables() const { return primary_input_size + auxiliary_input_size; }
};

class ProtoboardT {
public:
    ConstraintSystemT constraint_system;
    std::vector<FieldT> values{FieldT::one()};               // index 0 = ONE
    ProtoboardT() = default;
    ProtoboardT(const ProtoboardT &) = delete;
    ProtoboardT &operator=(const ProtoboardT &) = delete;

    uint32_t allocate_var_index(const std::string & = "") {
        constraint_system.auxiliary_input_size++;
        values.push_back(FieldT::zero());
        return (uint32_t)values.size() - 1;
    }
    FieldT &val(const VariableT &v) { return values.at(v.index); }
    const FieldT &val(const VariableT &v) const { return values.at(v.index); }
    FieldT lc_val(const LinearCombinationT &lc) const {
        FieldT s;
        for (const auto &t : lc.getTerms()) s += t.coeff * values.at(t.index);
        return s;
    }
    void add_r1cs_constraint(const ConstraintT &c, const std::string & = "") { constraint_system.constraints.emplace_back(new ConstraintT(c)); }
    // the first n allocated variables are the public inputs
    void set_input_sizes(size_t n) {
        const size_t total = constraint_system.num_variables();
        if (n > total) throw std::invalid_argument("set_input_sizes: more inputs than variables");
        constraint_system.primary_input_size = n;
        constraint_system.auxiliary_input_size = total - n;
    }
    size_t num_constraints() const { return constraint_system.num_constraints(); }
    size_t num_inputs() const { return constraint_system.num_inputs(); }
    size_t num_variables() const { return constraint_system.num_variables(); }
    std::vector<FieldT> primary_input() const { return std::vector<FieldT>(values.begin() + 1, values.begin() + 1 + (long)num_inputs()); }
    bool is_satisfied() const {
        for (const auto &c : constraint_system.constraints)
            if (lc_val(c->getA()) * lc_val(c->getB()) != lc_val(c->getC())) return false;
        return true;
    }
};
inline void VariableT::allocate(ProtoboardT &pb, const std::string &annotation) { index = pb.allocate_var_index(annotation); }

class VariableArrayT : public std::vector<VariableT> {
public:
    using std::vector<VariableT>::vector;
    void allocate(ProtoboardT &pb, size_t n, const std::string &annotation = "") {
        clear();
        for (size_t i = 0; i < n; i++) { VariableT v; v.allocate(pb, annotation); push_back(v); }
    }
    void fill_with_field_elements(ProtoboardT &pb, const std::vector<FieldT> &vals) const {
        for (size_t i = 0; i < vals.size() && i < size(); i++) pb.val((*this)[i]) = vals[i];
    }
    // little-endian bits of `v` into the first size() variables
    void fill_with_bits_of_ulong(ProtoboardT &pb, unsigned long v) const { for (size_t i = 0; i < size(); i++) pb.val((*this)[i]) = FieldT((long)((v >> i) & 1)); }
};

class GadgetT {
public:
    ProtoboardT &pb;
    const std::string annotation_prefix;
    GadgetT(ProtoboardT &in_pb, const std::string &in_annotation_prefix = "") : pb(in_pb), annotation_prefix(in_annotation_prefix) {}
};

inline VariableT make_variable(ProtoboardT &pb, const std::string &annotation = "") { VariableT v; v.allocate(pb, annotation); return v; }
inline VariableT make_variable(ProtoboardT &pb, const FieldT &value, const std::string &annotation = "") { VariableT v = make_variable(pb, annotation); pb.val(v) = value; return v; }
inline VariableArrayT make_var_array(ProtoboardT &pb, size_t n, const std::string &annotation = "") { VariableArrayT a; a.allocate(pb, n, annotation); return a; }

inline std::string FMT(const std::string &prefix, const char *fmt, ...) {
    char buf[256];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    return prefix + buf;
}

struct ppT { static void init_public_params() {} };           // no global state to set up here

// ------------------------------------------------------------------------------------------ dumps (src/export.cpp:157-221)
inline std::string r1cs2json(const ProtoboardT &pb) {
    const auto &cs = pb.constraint_system;
    auto lc = [](const LinearCombinationT &v) {
        std::string s = "{"; bool first = true;
        for (const auto &t : v.getTerms()) { if (!first) s += ","; first = false; s += "\"" + std::to_string(t.index) + "\": \"" + t.coeff.to_decimal() + "\""; }
        return s + "}";
    };
    std::string s = "{\n \"nPubInputs\": " + std::to_string(cs.num_inputs()) + ",\n \"nOutputs\": 0,\n \"nVars\": " + std::to_string(cs.num_variables() + 1) +
                    ",\n \"nConstraints\": " + std::to_string(cs.num_constraints()) + ",\n \"constraints\": [\n";
    for (size_t c = 0; c < cs.num_constraints(); c++) {
        const auto &k = *cs.constraints[c];
        s += "  [" + lc(k.getA()) + "," + lc(k.getB()) + "," + lc(k.getC()) + (c + 1 == cs.num_constraints() ? "]\n" : "],\n");
    }
    return s + " ]\n}";
}
inline std::string witness2json(const ProtoboardT &pb) {
    std::string s = "[\n";
    for (size_t i = 0; i < pb.values.size(); i++) s += " \"" + pb.values[i].to_decimal() + "\"" + (i + 1 == pb.values.size() ? "\n" : ",\n");
    return s + "]";
}

}  // namespace ethsnarks
