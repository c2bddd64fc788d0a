// ethsnarks_hip/stubs.hpp -- header-only C++ adapter: the reference's proving API on top of the C ABI of
// libzkhip.so (include/zkhip.h).  It re-exports, with the reference's own names and signatures,
//
//   ethsnarks::load_proving_key(const char*)                      src/stubs.hpp:18,  src/stubs.cpp:36-39
//   ethsnarks::get_domain(pb, pk, config)                         src/stubs.hpp:21,  src/stubs.cpp:61-75
//   ethsnarks::prove(ProverContextT&, ProtoboardT&)               src/stubs.hpp:19,  src/stubs.cpp:42-47
//   ethsnarks::stub_genkeys_from_pb(pb, pk_file, vk_file)         src/stubs.hpp:16,  src/stubs.cpp:77-87
//   ethsnarks::stub_prove_from_pb(pb, pk_raw)                     upstream wrapper, src/pinocchio/main.cpp:10,41
//   ethsnarks::stub_main_prove<GadgetT>(prog, argc, argv)         CLI helper shaped like stub_main_genkeys, src/stubs.hpp:36-55
//   ethsnarks::stub_genkeys<GadgetT>, stub_main_genkeys<GadgetT>  src/stubs.hpp:23-55
//   ethsnarks::stub_main_verify(prog, argc, argv)                 src/stubs.hpp:12,  src/stubs.cpp:90-132
//   ethsnarks::stub_verify(vk_json, proof_json)                   src/stubs.hpp:10,  src/stubs.cpp:16-33
//   ethsnarks::stub_test_proof_verify(pb)                         src/stubs.hpp:14,  src/stubs.cpp:135-148 (context fully initialised)
//
// so that a gadget binary written against ethsnarks keeps its source: it includes this header instead of
// "stubs.hpp" and links libzkhip.so instead of libsnark's prover.  The libsnark *front end* (protoboard,
// gadgets, r1cs_constraint_system) stays what it was; only the proving key container, the prover context
// and the prover itself are replaced.  Where the libsnark/libff headers of the ethsnarks build are on the include
// path they are used; where they are not (the reference checkout itself ships them as empty submodules) the
// stand-alone front end of ethsnarks_hip/circuit.hpp supplies the same names, so the adapter -- and gadget
// code written against that surface, ethsnarks_hip/gadgets.hpp -- compiles and runs either way.
//
// The adapter flattens pb.constraint_system once per context through the accessor chain the reference's
// own dumper uses (src/export.cpp:157-190: constraints[c]->getA().getTerms() -> {index, getCoeff()}),
// and passes pb.values (contiguous Fr[V+1], Montgomery, ONE at index 0) by pointer.
#pragma once
#include <zkhip.h>

#include <cstdint>
#include <cstdio>
#include <iterator>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <unistd.h>

#include <algorithm>
#include <ostream>
#include <type_traits>
#include <utility>

// ---- libsnark::Config, source-compatible with src/prover_config.hpp:8-85 (same members and defaults, the printer, and
// get_cpu_ranges).  The guard is the reference's own: whichever of the two headers comes first defines the struct, the
// other is skipped.  The CPU-cache knobs are accepted and ignored; multi_exp_c is honoured.
#ifndef ETHSNARKS_PROVER_CONFIG_HPP_
#define ETHSNARKS_PROVER_CONFIG_HPP_
namespace libsnark {
struct Config {
    Config() : num_threads(1), smt(false), fft("recursive"), swapAB(true), multi_exp_c(0), multi_exp_prefetch_locality(0),
               prefetch_stride(128), multi_exp_look_ahead(1) {}
    unsigned int num_threads;
    bool smt;
    std::string fft;
    std::vector<unsigned int> radixes;
    bool swapAB;
    unsigned int multi_exp_c;
    unsigned int multi_exp_prefetch_locality;
    unsigned int prefetch_stride;
    unsigned int multi_exp_look_ahead;
};
inline std::ostream &operator<<(std::ostream &os, const Config &c) {       // the reference's line, field for field
    os << "num_threads: " << c.num_threads << ", smt: " << c.smt << ", fft: " << c.fft << ", radixes: [";
    for (size_t i = 0; i < c.radixes.size(); i++) os << (i ? "," : "") << c.radixes[i];
    return os << "], exp_c: " << c.multi_exp_c << ", pre_stride: " << c.prefetch_stride << ", exp_preloc: "
              << c.multi_exp_prefetch_locality << ", exp_lookahead: " << c.multi_exp_look_ahead;
}
// [startIdx, length) cut into at most num_threads near-equal ranges (0 = one range here: no OpenMP on this side)
inline std::vector<std::pair<unsigned int, unsigned int>> get_cpu_ranges(unsigned int startIdx, unsigned int length, unsigned int num_threads = 0) {
    std::vector<std::pair<unsigned int, unsigned int>> ranges;
    if (startIdx >= length) return ranges;
    const unsigned int dist = length - startIdx, n = std::min<unsigned int>(num_threads ? num_threads : 1u, dist);
    const unsigned int chunk = dist / n, rem = dist % n;
    for (unsigned int i = 0; i + 1 < n; i++) { const unsigned int end = startIdx + chunk + (i < rem ? 1 : 0); ranges.emplace_back(startIdx, end); startIdx = end; }
    ranges.emplace_back(startIdx, length);
    return ranges;
}
}  // namespace libsnark
#endif

// ---- front-end types.  With libsnark on the include path its protoboard / gadget classes are used and the aliases of
// src/ethsnarks.hpp:31-41 are restated here -- NOT taken from that header, whose ProvingKeyT / ProverContextT (:43,48) name
// libsnark's CPU prover types and would collide with the ones below.  Its include guard is claimed, so gadget headers that
// say #include "ethsnarks.hpp" get these aliases instead.  Without libsnark, ethsnarks_hip/circuit.hpp supplies the names.
#if __has_include(<libsnark/gadgetlib1/protoboard.hpp>)
#ifdef ETHSNARKS_HPP_
#error "include ethsnarks_hip/stubs.hpp INSTEAD of the reference's ethsnarks.hpp / stubs.hpp (their ProvingKeyT / ProverContextT are libsnark's CPU prover types)"
#endif
#define ETHSNARKS_HPP_
#include <libsnark/gadgetlib1/protoboard.hpp>
#include <libsnark/gadgetlib1/gadget.hpp>
#if defined(CURVE_MCL_BN128)
#include <libff/algebra/curves/mcl_bn128/mcl_bn128_pp.hpp>
#else
#include <libff/algebra/curves/alt_bn128/alt_bn128_pp.hpp>
#endif
namespace ethsnarks {
#if defined(CURVE_MCL_BN128)
typedef libff::mcl_bn128_pp ppT;            // its Fr wrapper's memory image is defined in the absent libff fork: values cross the
#define ETHSNARKS_HIP_FR_VIA_BIGINT 1       // boundary through as_bigint() (canonical), not by reinterpreting the object
#else
typedef libff::alt_bn128_pp ppT;            // Fr = libff::Fp_model<4>: 4 x u64 Montgomery limbs, passed by pointer
#endif
typedef libff::Fr<ppT> FieldT;
typedef libsnark::r1cs_constraint<FieldT> ConstraintT;
typedef libsnark::protoboard<FieldT> ProtoboardT;
typedef libsnark::pb_variable<FieldT> VariableT;
typedef libsnark::pb_variable_array<FieldT> VariableArrayT;
typedef libsnark::pb_linear_combination<FieldT> LinearCombinationT;
typedef libsnark::pb_linear_combination_array<FieldT> LinearCombinationArrayT;
typedef libsnark::linear_term<FieldT> LinearTermT;
typedef libsnark::gadget<FieldT> GadgetT;
}  // namespace ethsnarks
#define ETHSNARKS_HIP_HAVE_LIBSNARK 1
#else
#include "circuit.hpp"            // stand-alone front end with the same surface (FieldT, ProtoboardT, VariableT, ...)
#define ETHSNARKS_HIP_HAVE_LIBSNARK 0
#endif

namespace ethsnarks {

// HIP device the contexts of this process are created on (the reference has no such notion; default 0)
inline unsigned int &hip_device() { static unsigned int d = 0; return d; }

struct zk_error : std::runtime_error {
    int code;
    zk_error(int c) : std::runtime_error(std::string(zk_strerror(c)) + ": " + zk_last_error()), code(c) {}
};
inline void zk_check(int rc) { if (rc != ZK_OK) throw zk_error(rc); }

// stub_verify (src/stubs.cpp:16-33): needs no libsnark types at all
inline bool stub_verify(const char *vk_json, const char *proof_json) {
    int ok = 0;
    zk_check(zk_verify(vk_json, proof_json, &ok));      // malformed JSON throws, as the reference's parser does
    return ok == 1;
}

// ProvingKeyT: owning handle of the nozk proving key (r1cs_gg_ppzksnark_zok_proving_key_nozk, hpp:171-274)
class ProvingKeyT {
public:
    ProvingKeyT() = default;
    explicit ProvingKeyT(zk_pk *h) : h_(h, zk_pk_free) {}
    const zk_pk *get() const { return h_.get(); }
private:
    std::shared_ptr<zk_pk> h_;
};

inline ProvingKeyT load_proving_key(const char *pk_file) {
    zk_pk *h = nullptr;
#if defined(CURVE_MCL_BN128)
    zk_check(zk_pk_load_raw(pk_file, ZK_CODEC_MCL_BN128, &h));     // inferred element layout, parity unpinned (include/zkhip.h)
#else
    zk_check(zk_pk_load_raw(pk_file, ZK_CODEC_ALT_BN128, &h));     // the reference asserts on a missing file (utils.hpp:180)
#endif
    return ProvingKeyT(h);
}

// pk_bellman2ethsnarks (src/export.hpp:24, src/export.cpp:267-328): bellman-style JSON key -> nozk `.raw`.
// The reference returns true even when the input cannot be opened (it only prints); here that is `false`.
inline bool pk_bellman2ethsnarks(const std::string &bellman_pk_file, const std::string &pk_file) {
    const int rc = zk_pk_bellman2ethsnarks(bellman_pk_file.c_str(), pk_file.c_str());
    if (rc != ZK_OK) std::cerr << "pk_bellman2ethsnarks: " << zk_strerror(rc) << ": " << zk_last_error() << std::endl;
    return rc == ZK_OK;
}

// the "domain" of the reference is an evaluation_domain object; here it is a property of the context.
struct DomainT { uint32_t m; };

// ProverContextT (hpp:279-291), member for member: borrows the key and (optionally) a constraint system, carries the
// config and the domain, owns scratch.  The reference's scratch vectors stay empty (the scratch lives in HBM, owned by
// `ctx`); they are here so that code which sizes or clears them keeps compiling.  One context per concurrent prover.
typedef std::remove_reference<decltype(std::declval<ProtoboardT &>().constraint_system)>::type ConstraintSystemT_;
struct ProverContextT {
    ProvingKeyT &provingKey;
    const ConstraintSystemT_ *constraint_system = nullptr;       // when set it is the system that is proven (else pb.constraint_system)
    libsnark::Config config;
    std::shared_ptr<DomainT> domain;
    std::vector<decltype(std::declval<const FieldT &>().as_bigint())> scratch_exponents;
    std::vector<FieldT> aA, aB, aH;
    std::shared_ptr<zk_ctx> ctx;                                  // added: the device-side context, built on first use
    explicit ProverContextT(ProvingKeyT &pk) : provingKey(pk) {}
};

namespace detail {
struct Flat { std::vector<uint32_t> ptr, col; std::vector<uint64_t> coeff; };
// Montgomery limbs of an Fr element for the C ABI
inline void push_limbs(std::vector<uint64_t> &dst, const FieldT &c) {
#ifdef ETHSNARKS_HIP_FR_VIA_BIGINT
    const auto b = c.as_bigint();                                   // canonical; converted in bulk by zk_fr_convert below
    dst.insert(dst.end(), b.data, b.data + 4);
#else
    static_assert(sizeof(FieldT) == 32, "FieldT must be 4 x u64 Montgomery limbs (libff::Fp_model<4>)");
    const uint64_t *limbs = reinterpret_cast<const uint64_t *>(&c);
    dst.insert(dst.end(), limbs, limbs + 4);
#endif
}
template <class LC> void push_row(Flat &f, const LC &lc) {
    for (const auto &t : lc.getTerms()) {                          // src/export.cpp:157-171
        f.col.push_back((uint32_t)t.index);
        push_limbs(f.coeff, t.getCoeff());
    }
    f.ptr.push_back((uint32_t)f.col.size());
}
struct FlatSystem {
    Flat A, B, C; uint32_t nC, nIn, V;
    template <class CS> explicit FlatSystem(const CS &cs) : nC((uint32_t)cs.num_constraints()), nIn((uint32_t)cs.num_inputs()), V((uint32_t)cs.num_variables()) {
        A.ptr = B.ptr = C.ptr = {0};
        for (size_t c = 0; c < cs.num_constraints(); c++) {        // src/export.cpp:183-190
            push_row(A, cs.constraints[c]->getA()); push_row(B, cs.constraints[c]->getB()); push_row(C, cs.constraints[c]->getC());
        }
#ifdef ETHSNARKS_HIP_FR_VIA_BIGINT
        for (Flat *f : {&A, &B, &C}) zk_check(zk_fr_convert(f->coeff.data(), (uint32_t)(f->coeff.size() / 4), 1));
#endif
    }
    zk_csr a() const { return zk_csr{nC, A.ptr.data(), A.col.data(), A.coeff.data()}; }
    zk_csr b() const { return zk_csr{nC, B.ptr.data(), B.col.data(), B.coeff.data()}; }
    zk_csr c() const { return zk_csr{nC, C.ptr.data(), C.col.data(), C.coeff.data()}; }
};
inline void ensure_context(ProverContextT &context, const ProtoboardT &pb) {
    if (context.ctx) return;
    const FlatSystem f(context.constraint_system ? *context.constraint_system : pb.constraint_system);
    const zk_csr a = f.a(), b = f.b(), c = f.c();
    zk_config cfg{context.config.multi_exp_c, hip_device(), 0, 1, 1, ZK_SCHED_LATENCY};      // prove() is one synchronous proof at a time
    zk_ctx *h = nullptr;
    zk_check(zk_ctx_create_sized(context.provingKey.get(), &a, &b, &c, f.nC, f.nIn, f.V, &cfg, sizeof(cfg), &h));   // the size this translation unit was compiled with
    context.ctx.reset(h, zk_ctx_destroy);
}
// pb.values (Fr[V + 1], ONE at index 0) as the C ABI wants it
struct Witness {
    std::vector<uint64_t> copy; const uint64_t *ptr; int canonical;
    explicit Witness(const ProtoboardT &pb) {
#ifdef ETHSNARKS_HIP_FR_VIA_BIGINT
        copy.reserve(4 * pb.values.size());
        for (const auto &v : pb.values) push_limbs(copy, v);
        ptr = copy.data(); canonical = 1;
#else
        ptr = reinterpret_cast<const uint64_t *>(pb.values.data()); canonical = 0;
#endif
    }
};
}  // namespace detail

// get_domain (src/stubs.cpp:61-75): same size rule; building the context uploads key + CSR and the twiddles
inline const std::shared_ptr<DomainT> get_domain(ProtoboardT &pb, const ProvingKeyT &, const libsnark::Config &) {
    const auto &cs = pb.constraint_system;
    return std::make_shared<DomainT>(DomainT{zk_domain_size((uint32_t)cs.num_constraints(), (uint32_t)cs.num_inputs())});
}

// The reference's prover reports its phases through libff::enter_block / leave_block (tcc:454-544) unless
// libff::inhibit_profiling_info is set.  Here the same block names are printed after the proof, with the GPU time of each
// phase (zk_timings), when this flag is cleared -- it starts out set: the phases overlap on the device, so the lines are a
// report, not a nesting trace, and a drop-in caller's stdout stays what it was.  ZK_PROFILING_INFO=1 in the environment
// clears it too (read once).
inline bool &inhibit_profiling_info() {
    static bool inhibit = !(std::getenv("ZK_PROFILING_INFO") && std::getenv("ZK_PROFILING_INFO")[0] == '1');
    return inhibit;
}
namespace detail {
inline void print_prover_blocks(const zk_timings &t, std::FILE *f = stdout) {
    const double total = (double)t.h2d_witness + t.gpu_total + t.host_finish;
    auto leave = [&](int indent, const char *name, double ms) { std::fprintf(f, "%*s(leave) %-46s [%0.4fs]\n", 2 * indent, "", name, ms * 1e-3); };
    std::fprintf(f, "(enter) Call to r1cs_gg_ppzksnark_zok_prover\n");
    leave(1, "Compute the polynomial H", t.compute_h);                              // tcc:460-475
    std::fprintf(f, "  (enter) Compute the proof\n");                              // tcc:485
    leave(2, "Compute evaluation to A-query", t.a_query);                           // tcc:487-496
    leave(2, "Compute evaluation to B-query", t.b_query);                           // tcc:498-507
    leave(2, "Compute evaluation to H-query", t.h_query);                           // tcc:509-519
    leave(2, "Compute evaluation to L-query", t.l_query);                           // tcc:521-531
    // what is left of the call after "Compute the polynomial H" (the phases overlap on the GPU: the block cannot exceed the call minus H)
    const double after_h = (double)t.gpu_total - t.compute_h;
    leave(1, "Compute the proof", (after_h > 0 ? after_h : 0.0) + t.host_finish);   // tcc:542
    leave(0, "Call to r1cs_gg_ppzksnark_zok_prover", total);                        // tcc:544
    std::fprintf(f, "* G1 elements in proof: 2\n* G2 elements in proof: 1\n* Proof size in bits: %d\n", 2 * 256 + 512);   // proof.print_size(), tcc:547 (uncompressed affine coordinates here)
}
}  // namespace detail

// prove (src/stubs.cpp:42-47): proof JSON of src/export.cpp:99-121
namespace detail {
inline std::string prove_const(ProverContextT &context, const ProtoboardT &pb) {
    ensure_context(context, pb);
    const Witness w(pb);                                            // ONE at index 0 (tcc:492-493)
    zk_proof proof;
    if (inhibit_profiling_info()) zk_check(zk_prove(context.ctx.get(), w.ptr, w.canonical, &proof));
    else { zk_timings t; zk_check(zk_prove_timed(context.ctx.get(), w.ptr, w.canonical, &proof, &t)); print_prover_blocks(t); }
    const uint32_t nIn = (uint32_t)pb.constraint_system.num_inputs();
    size_t len = 0;
    zk_proof_to_json(&proof, w.ptr + 4, nIn, w.canonical, nullptr, 0, &len);
    std::string out(len + 1, '\0');
    zk_check(zk_proof_to_json(&proof, w.ptr + 4, nIn, w.canonical, &out[0], out.size(), &len));
    out.resize(len);
    return out;
}
}  // namespace detail
inline std::string prove(ProverContextT &context, ProtoboardT &pb) { return detail::prove_const(context, pb); }

// Throughput form of prove() (no reference counterpart: the reference's ProverContext is one synchronous prover per thread,
// hpp:279-291).  N contexts of ONE key and ONE constraint system stay in flight on the GPU; submit() hands over the values of a
// protoboard of that system, next() returns the proofs in submission order, each the JSON prove() would return.  A context that
// is still proving takes its next witness ahead (zk_prove_stage: copied to the device under the running proof), so up to 2 N
// submissions are accepted before next() has to be called.  Single-threaded like the contexts themselves.
class ProverPipeline {
 public:
    ProverPipeline(const ProvingKeyT &pk, const ProtoboardT &pb, unsigned contexts = 3, const libsnark::Config &config = libsnark::Config())
        : nIn_((uint32_t)pb.constraint_system.num_inputs()), n_values_(pb.values.size()) {
        const detail::FlatSystem f(pb.constraint_system);
        const zk_csr a = f.a(), b = f.b(), c = f.c();
        zk_config cfg{config.multi_exp_c, hip_device(), 0, 1, 1, ZK_SCHED_OVERLAP};
        for (unsigned i = 0; i < (contexts ? contexts : 1); i++) {
            zk_ctx *h = nullptr;
            zk_check(zk_ctx_create_sized(pk.get(), &a, &b, &c, f.nC, f.nIn, f.V, &cfg, sizeof(cfg), &h));
            slots_.emplace_back();
            slots_.back().ctx.reset(h, zk_ctx_destroy);
        }
    }
    size_t pending() const { return submitted_ - collected_; }
    bool full() const { return pending() >= 2 * slots_.size(); }
    void submit(const ProtoboardT &pb) {
        if (pb.values.size() != n_values_) throw std::invalid_argument("ProverPipeline: protoboard of another constraint system");
        if (full()) throw std::runtime_error("ProverPipeline: full (call next() first)");
        Slot &s = slots_[submitted_ % slots_.size()];
        const detail::Witness w(pb);
        std::vector<uint64_t> inputs(w.ptr + 4, w.ptr + 4 + 4 * (size_t)nIn_);
        if (!s.busy) {
            zk_check(zk_prove_submit(s.ctx.get(), w.ptr, w.canonical));
            s.busy = true; s.inputs = std::move(inputs); s.canonical = w.canonical;
        } else {                                                   // second round: this context's next witness goes ahead of its turn
            zk_check(zk_prove_stage(s.ctx.get(), w.ptr, 1, w.canonical));
            s.staged = true; s.staged_inputs = std::move(inputs); s.staged_canonical = w.canonical;
        }
        submitted_++;
    }
    std::string next() {
        if (!pending()) throw std::runtime_error("ProverPipeline: nothing submitted");
        Slot &s = slots_[collected_ % slots_.size()];
        zk_partials part; zk_proof proof;
        zk_check(zk_prove_collect(s.ctx.get(), &part, nullptr));
        zk_check(zk_prove_combine(s.ctx.get(), &part, 1, &proof));
        size_t len = 0;
        zk_proof_to_json(&proof, s.inputs.data(), nIn_, s.canonical, nullptr, 0, &len);
        std::string out(len + 1, '\0');
        zk_check(zk_proof_to_json(&proof, s.inputs.data(), nIn_, s.canonical, &out[0], out.size(), &len));
        out.resize(len);
        s.busy = false;
        if (s.staged) {
            zk_check(zk_prove_submit_staged(s.ctx.get()));
            s.busy = true; s.staged = false; s.inputs = std::move(s.staged_inputs); s.canonical = s.staged_canonical;
        }
        collected_++;
        return out;
    }
 private:
    struct Slot { std::shared_ptr<zk_ctx> ctx; bool busy = false, staged = false; std::vector<uint64_t> inputs, staged_inputs; int canonical = 0, staged_canonical = 0; };
    std::vector<Slot> slots_;
    uint32_t nIn_; size_t n_values_;
    size_t submitted_ = 0, collected_ = 0;
};

namespace detail {
inline std::string prove_from_pb_const(const ProtoboardT &pb, const char *pk_raw) {
    ProvingKeyT pk = load_proving_key(pk_raw);
    ProverContextT context(pk);
    context.constraint_system = &pb.constraint_system;
    context.config = libsnark::Config();
    context.domain = std::make_shared<DomainT>(DomainT{zk_domain_size((uint32_t)pb.constraint_system.num_constraints(), (uint32_t)pb.constraint_system.num_inputs())});
    return prove_const(context, pb);
}
inline int genkeys_from_pb_const(const ProtoboardT &pb, const char *pk_file, const char *vk_file);
}  // namespace detail
inline std::string stub_prove_from_pb(ProtoboardT &pb, const char *pk_raw) { return detail::prove_from_pb_const(pb, pk_raw); }   // src/pinocchio/main.cpp:41

inline int stub_genkeys_from_pb(ProtoboardT &pb, const char *pk_file, const char *vk_file) { return detail::genkeys_from_pb_const(pb, pk_file, vk_file); }   // src/stubs.cpp:77-87
inline int detail::genkeys_from_pb_const(const ProtoboardT &pb, const char *pk_file, const char *vk_file) {
    const detail::FlatSystem f(pb.constraint_system);
    const zk_csr a = f.a(), b = f.b(), c = f.c();
    uint64_t toxic[20];
    for (int i = 0; i < 5; i++) {                                   // t, alpha, beta, gamma, delta (tcc:283-287)
        const auto v = FieldT::random_element().as_bigint();
        std::memcpy(toxic + 4 * i, v.data, 32);
    }
    zk_pk *pk = nullptr; zk_vk *vk = nullptr;
    if (zk_keygen(&a, &b, &c, f.nC, f.nIn, f.V, toxic, (int)hip_device(), &pk, &vk) != ZK_OK) return 1;
    size_t len = 0;
    zk_vk_to_json(vk, nullptr, 0, &len);
    std::string js(len + 1, '\0');
    int rc = zk_vk_to_json(vk, &js[0], js.size(), &len);
    if (rc == ZK_OK) { std::ofstream fh(vk_file, std::ios::binary); fh.write(js.data(), (std::streamsize)len); rc = fh ? ZK_OK : ZK_ERR_IO; }
#if defined(CURVE_MCL_BN128)
    if (rc == ZK_OK) rc = zk_pk_save_raw(pk, pk_file, ZK_CODEC_MCL_BN128);
#else
    if (rc == ZK_OK) rc = zk_pk_save_raw(pk, pk_file, ZK_CODEC_ALT_BN128);
#endif
    zk_pk_free(pk); zk_vk_free(vk);
    return rc == ZK_OK ? 0 : 1;
}

// stub_test_proof_verify (src/stubs.hpp:14, src/stubs.cpp:135-148): keygen -> prove -> verify in memory.  Unlike the
// reference it sets up the whole context (the reference leaves constraint_system and domain unset).
inline bool stub_test_proof_verify(const ProtoboardT &pb) {
    const char *tmpdir = std::getenv("TMPDIR");
    std::string tmpl = std::string(tmpdir && *tmpdir ? tmpdir : "/tmp") + "/zkhip_pk_XXXXXX";
    const int fd = mkstemp(&tmpl[0]);
    if (fd < 0) return false;
    close(fd);
    const std::string pk_tmp = tmpl, vk_tmp = pk_tmp + ".vk.json";
    if (detail::genkeys_from_pb_const(pb, pk_tmp.c_str(), vk_tmp.c_str()) != 0) return false;
    std::ifstream vf(vk_tmp, std::ios::binary);
    const std::string vk((std::istreambuf_iterator<char>(vf)), std::istreambuf_iterator<char>());
    const std::string proof = detail::prove_from_pb_const(pb, pk_tmp.c_str());
    std::remove(pk_tmp.c_str()); std::remove(vk_tmp.c_str());
    return stub_verify(vk.c_str(), proof.c_str());
}

// stub_genkeys<GadgetT> / stub_main_genkeys<GadgetT> (src/stubs.hpp:23-55): build the gadget's constraints on a fresh
// protoboard and write the key pair; argv = {command, pk-output.raw, vk-output.json}; 0 on success, 1 otherwise
template <class GadgetT>
int stub_genkeys(const char *pk_file, const char *vk_file) {
    ppT::init_public_params();
    ProtoboardT pb;
    GadgetT mod(pb, "module");
    mod.generate_r1cs_constraints();
    return stub_genkeys_from_pb(pb, pk_file, vk_file);
}
template <class GadgetT>
int stub_main_genkeys(const char *prog_name, int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "Usage: " << prog_name << " " << argv[0] << " <pk-output.raw> <vk-output.json>" << std::endl;
        return 1;
    }
    if (stub_genkeys<GadgetT>(argv[1], argv[2]) != 0) {
        std::cerr << "Error: failed to generate proving and verifying keys" << std::endl;
        return 1;
    }
    return 0;
}

// stub_main_verify (src/stubs.cpp:90-132): argv = {command, vk.json, proof.json}; 0 verified, 1 usage / rejected,
// 2 a file cannot be opened
inline int stub_main_verify(const char *prog_name, int argc, const char **argv) {
    if (argc < 3) {
        std::cerr << "Usage: " << prog_name << " " << argv[0] << " <vk.json> <proof.json>" << std::endl;
        return 1;
    }
    std::string text[2];
    for (int i = 0; i < 2; i++) {
        std::ifstream in(argv[1 + i], std::ios::binary);
        if (!in) { std::cerr << "Error: cannot open " << argv[1 + i] << std::endl; return 2; }
        text[i].assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
    }
    bool ok = false;
    try { ok = stub_verify(text[0].c_str(), text[1].c_str()); } catch (const std::exception &e) { std::cerr << "Error: " << e.what() << std::endl; }
    if (ok) return 0;
    std::cerr << "Error: failed to verify proof!" << std::endl;
    return 1;
}

// stub_main_prove: argv helper shaped like stub_main_genkeys (src/stubs.hpp:36-55)
template <class GadgetT>
int stub_main_prove(const char *prog_name, int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "Usage: " << prog_name << " " << argv[0] << " <pk-input.raw> <proof-output.json>" << std::endl;
        return 1;
    }
    ppT::init_public_params();
    ProtoboardT pb;
    GadgetT mod(pb, "module");
    mod.generate_r1cs_constraints();
    mod.generate_r1cs_witness();
    try {
        std::ofstream fh(argv[2], std::ios::binary);
        fh << stub_prove_from_pb(pb, argv[1]);
        return fh ? 0 : 1;
    } catch (const std::exception &e) {
        std::cerr << "Error: failed to prove: " << e.what() << std::endl;
        return 1;
    }
}
}  // namespace ethsnarks
