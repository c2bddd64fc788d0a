/*
 * zkhip.h -- C ABI of libzkhip.so, the MI355X (gfx950) Groth16 proving backend for ethsnarks.
 *
 * Drop-in boundary for the reference's proving path (citations into zkh2018/ethsnarks):
 *
 *   zk_pk_load_raw      <- ethsnarks::load_proving_key(pk_file)              src/stubs.cpp:36-39
 *                          loadFromFile<ProvingKeyT>                         src/utils.hpp:176-185
 *                          operator>>(istream&, pk_nozk&)                    r1cs_gg_ppzksnark_zok.tcc:124-143
 *   zk_pk_save_raw      <- writeToFile<ProvingKeyT> / operator<<             src/utils.hpp:166-173, tcc:108-122
 *   zk_pk_from_bellman_json / zk_pk_bellman2ethsnarks <- pk_bellman2ethsnarks + readG1/readG2   src/export.cpp:223-328
 *   zk_pk_alt2mcl / zk_pk_mcl2nozk <- pk_alt2mcl / pk_mcl2nozk                src/export.cpp:330-408
 *   zk_pk_from_parts    <- r1cs_gg_ppzksnark_zok_proving_key_nozk ctor       r1cs_gg_ppzksnark_zok.hpp:171-233
 *   zk_ctx_create       <- ProverContext<ppT>(pk) + get_domain(pb, pk, cfg)  hpp:279-291, src/stubs.cpp:61-75
 *   zk_prove            <- r1cs_gg_ppzksnark_zok_prover(ctx, pb.values)      tcc:451-550 (via prove(), stubs.cpp:42-47)
 *   zk_proof_to_json    <- proof_to_json(proof, primary_input)               src/export.cpp:99-121
 *   zk_config           <- libsnark::Config                                   src/prover_config.hpp:8-35
 *   zk_keygen           <- r1cs_gg_ppzksnark_zok_generator + nozk conversion  tcc:277-449, hpp:209-233
 *                          (stub_genkeys_from_pb, src/stubs.cpp:77-87)
 *   zk_vk_to_json       <- vk2json                                            src/export.cpp:124-145
 *   zk_vk_from_json / zk_proof_from_json <- vk_from_json / proof_from_json   src/import.cpp:161-223
 *   zk_verify / ethsnarks_verify <- stub_verify / ethsnarks_verify          src/stubs.cpp:16-33, src/verify_dll.cpp:3-10
 *
 * Plain C types only.  Field elements are 4 x u64 little-endian limbs; "Montgomery" means the
 * libff::Fp_model<4> in-memory form (value * 2^256 mod p), which is what `pb.values` and the `.raw`
 * key file contain.  Affine G1 = {x, y} (8 u64), affine G2 = {x.c0, x.c1, y.c0, y.c1} (16 u64); an
 * all-zero point encodes infinity.  All functions return 0 (ZK_OK) or a positive error code; none
 * throws or aborts.  zk_last_error() gives a thread-local human-readable message.
 *
 * There is NO CPU implementation behind this ABI: every compute entry point needs a HIP device and
 * fails with ZK_ERR_NODEVICE / ZK_ERR_HIP otherwise.
 */
#ifndef ZKHIP_H
#define ZKHIP_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZK_OK 0
#define ZK_ERR_ARG 1       /* null / inconsistent argument */
#define ZK_ERR_IO 2        /* cannot open / read / write file (reference: assert in utils.hpp:180) */
#define ZK_ERR_FORMAT 3    /* malformed .raw stream */
#define ZK_ERR_HIP 4       /* HIP runtime error */
#define ZK_ERR_NOMEM 5
#define ZK_ERR_SHAPE 6     /* key does not match the constraint system (DEBUG asserts tcc:477-483) */
#define ZK_ERR_DEGREE 7    /* H has wrong degree: witness does not satisfy the R1CS (asserts tcc:472-474) */
#define ZK_ERR_NODEVICE 8
#define ZK_ERR_BUFFER 9    /* output buffer too small */
#define ZK_ERR_INTERNAL 10 /* an exception other than an allocation failure reached the C ABI's barrier (text in zk_last_error) */

#define ZK_CODEC_ALT_BN128 0   /* upstream libff alt_bn128 stream layout */
#define ZK_CODEC_MCL_BN128 1   /* the reference's default curve build (CMakeLists.txt:47-54); its element encoding lives in the
                                  absent libff fork and is INFERRED (same flag + raw Montgomery coordinates as ALT_BN128):
                                  parity unpinned -- the reference holds no key file (open risk).
                                  TODAY THE TWO CODEC VALUES READ AND WRITE BYTE-IDENTICAL FILES: the value is validated and
                                  recorded, nothing else; a verified mcl layout would change the point reader / writer only */

/* ABI version of this header: bumped whenever a struct below changes layout or a function changes signature.
 * 3: zk_config = {multi_exp_c, device, shard_rank, shard_count, max_batch, schedule} (24 bytes; rounds 1-2 had 16 / 24).
 * A client checks zk_abi_version() == ZK_ABI_VERSION once after loading the library, or passes the size of the zk_config
 * it was compiled with to zk_ctx_create_sized (members it does not know read as 0 = their defaults). */
#define ZK_ABI_VERSION 3

typedef struct zk_pk zk_pk;
typedef struct zk_vk zk_vk;
typedef struct zk_ctx zk_ctx;
typedef struct zk_wplan zk_wplan;

typedef struct {
    uint32_t n_rows;
    const uint32_t *row_ptr;   /* n_rows + 1 */
    const uint32_t *col;       /* variable index; 0 is the constant ONE */
    const uint64_t *coeff;     /* nnz x 4, Montgomery Fr */
} zk_csr;

/* mirrors libsnark::Config (src/prover_config.hpp:8-35).  CPU-cache knobs of the reference
 * (num_threads, smt, fft, radixes, prefetch_*, look_ahead) have no meaning on the GPU and are
 * accepted and ignored by the C++ adapter; what remains: */
typedef struct {
    uint32_t multi_exp_c;      /* Pippenger window bits, 0 = auto (Config::multi_exp_c) */
    uint32_t device;           /* HIP device ordinal */
    uint32_t shard_rank;       /* MSM base-range sharding: this context owns shard_rank of shard_count */
    uint32_t shard_count;      /* 0 or 1 = unsharded */
    uint32_t max_batch;        /* proofs of this circuit one launch sequence may carry (zk_prove_batch); 0 or 1 = one at a time */
    uint32_t schedule;         /* ZK_SCHED_OVERLAP (0): the proof's sorts, accumulations and reduction tails on five streams --
                                * lowest latency of one proof, what large circuits want; ZK_SCHED_ONE_STREAM (1): every launch on one
                                * stream -- for many small proofs in many contexts (a hardware queue runs its dispatches in order, so
                                * fewer streams per context leave more queues to other contexts: profiles/r02_small_circuit_concurrency.txt);
                                * ZK_SCHED_LATENCY (2): ZK_SCHED_OVERLAP for a context that proves ONE synchronous proof at a time (zk_prove =
                                * ethsnarks::prove): its streams share one priority, which shortens a lone proof of 2^16 constraints and more
                                * by 3-19 % and costs pipelined contexts 10 % (profiles/r03_queue_mapping.txt) */
} zk_config;
#define ZK_SCHED_OVERLAP 0
#define ZK_SCHED_ONE_STREAM 1
#define ZK_SCHED_LATENCY 2

/* canonical (non-Montgomery) affine coordinates; *_inf != 0 => point at infinity, printed as (0, 1) */
typedef struct {
    uint64_t a_x[4], a_y[4];
    uint64_t b_x_c0[4], b_x_c1[4], b_y_c0[4], b_y_c1[4];
    uint64_t c_x[4], c_y[4];
    uint32_t a_inf, b_inf, c_inf, _pad;
} zk_proof;

/* per-shard partial results of the four multi-exponentiations, XYZZ coordinates, Montgomery:
 * At, Ht, Lt: 4 x 4 u64 each (G1); Bt: 4 x 8 u64 (G2)  => 3*128 + 256 = 640 bytes.
 * The proof only ever uses Ht + Lt (C = Ht + Lt, tcc:540): when the H- and the L-query share their bucket set (the usual case)
 * ONE bucket reduction serves both, Ht carries the sum and Lt the point at infinity (all zero); zk_prove_combine adds them either way. */
typedef struct {
    uint64_t At[16], Bt[32], Ht[16], Lt[16];
} zk_partials;

/* phase times in milliseconds, named after the reference's enter_block labels (tcc:460-542) */
typedef struct {
    float h2d_witness;
    float compute_h;           /* "Compute the polynomial H" */
    float a_query, b_query, h_query, l_query;   /* "Compute the proof's A-query" ... "L-query": GPU time of each multi-exponentiation */
    float gpu_total;           /* first kernel to last copy */
    float host_finish;         /* "Compute the proof": final additions (tcc:533-540), affine conversion */
    float acc_a, acc_b, acc_h, acc_l;   /* k_msm_accumulate launch of each query, HIP events on its stream */
} zk_timings;

/* ---- library / device */
const char *zk_version(void);
uint32_t zk_abi_version(void);
const char *zk_strerror(int code);
const char *zk_last_error(void);
int zk_device_count(int *count);

/* ---- proving key (host object; zk_ctx_create uploads it) */
int zk_pk_load_raw(const char *path, int codec, zk_pk **out);
int zk_pk_save_raw(const zk_pk *pk, const char *path, int codec);
/* bellman / snarkjs style proving-key JSON (keys A, B1, B2, C, hExps, vk_alfa_1, vk_beta_1/2, vk_delta_1/2; Jacobian
 * decimal triples) -> nozk key, exactly as pk_bellman2ethsnarks maps it (one public input: L = C[2..]);
 * zk_pk_bellman2ethsnarks writes the `.raw` file the reference's converter writes */
int zk_pk_from_bellman_json(const char *json_path, zk_pk **out);
int zk_pk_bellman2ethsnarks(const char *bellman_pk_json, const char *pk_raw);
/* the reference's offline converters over the FULL (zero-knowledge) proving key stream (tcc:53-90):
 * pk_alt2mcl (src/export.cpp:352-397: every coordinate goes through its decimal string) and
 * pk_mcl2nozk (src/export.cpp:399-408: nozk conversion of hpp:209-233, written with the MCL codec) */
int zk_pk_alt2mcl(const char *alt_pk_file, const char *mcl_pk_file);
int zk_pk_mcl2nozk(const char *mcl_pk_file, const char *nozk_pk_file);
int zk_pk_from_parts(const uint64_t *alpha_g1, const uint64_t *beta_g1, const uint64_t *beta_g2,
                     const uint64_t *delta_g1, const uint64_t *delta_g2,
                     uint32_t a_domain, uint32_t nA, const uint32_t *a_idx, const uint64_t *a_val,
                     uint32_t b_domain, uint32_t nB, const uint32_t *b_idx, const uint64_t *b_val,
                     uint32_t nH, const uint64_t *H, uint32_t nL, const uint64_t *L, zk_pk **out);
/* sizes[0..5] = A.domain, nA, B.domain, nB, nH, nL */
int zk_pk_sizes(const zk_pk *pk, uint32_t sizes[6]);
/* which: 0 alpha_g1 1 beta_g1 2 beta_g2 3 delta_g1 4 delta_g2 5 A.idx 6 A.val 7 B.idx 8 B.val 9 H 10 L */
const void *zk_pk_part(const zk_pk *pk, int which);
void zk_pk_free(zk_pk *pk);

/* ---- key generation: the QAP is evaluated at t on the host, the fixed-base
 * exponentiations (A, B, H, L queries, gammaABC) run on the GPU.
 * toxic_canon = t, alpha, beta, gamma, delta as canonical 4 x u64 each (the reference draws them with
 * Fr::random_element(), tcc:283-287).  Generators: G1 (1, 2), G2 the standard alt_bn128 generator. */
int zk_keygen(const zk_csr *A, const zk_csr *B, const zk_csr *C, uint32_t nC, uint32_t nIn, uint32_t V,
              const uint64_t toxic_canon[20], int device, zk_pk **pk_out, zk_vk **vk_out);
int zk_vk_to_json(const zk_vk *vk, char *buf, size_t cap, size_t *len);
/* vk_from_json / proof_from_json (src/import.cpp:161-223): "0x" hex or decimal strings, Fq2 as [c1, c0].
 * zk_proof_from_json: the proof's public inputs come back canonical (4 x u64 each) in inputs_canon[0 .. *n_inputs);
 * ZK_ERR_BUFFER if cap is too small (*n_inputs still set), ZK_ERR_FORMAT on malformed text or coordinates >= q */
int zk_vk_from_json(const char *vk_json, zk_vk **out);
int zk_proof_from_json(const char *proof_json, zk_proof *out, uint64_t *inputs_canon, uint32_t cap, uint32_t *n_inputs);
void zk_vk_free(zk_vk *vk);

/* ---- prover context: uploads bases + CSR once, builds domain tables, owns all scratch.
 * Borrows nothing after return (pk and CSR may be freed), one context per concurrent prover. */
uint32_t zk_domain_size(uint32_t nC, uint32_t nIn);                      /* src/stubs.cpp:49-65 */
int zk_ctx_create(const zk_pk *pk, const zk_csr *A, const zk_csr *B, const zk_csr *C,
                  uint32_t nC, uint32_t nIn, uint32_t V, const zk_config *cfg, zk_ctx **out);
int zk_ctx_create_sized(const zk_pk *pk, const zk_csr *A, const zk_csr *B, const zk_csr *C,
                        uint32_t nC, uint32_t nIn, uint32_t V, const zk_config *cfg, size_t cfg_size, zk_ctx **out);
void zk_ctx_destroy(zk_ctx *ctx);

/* witness: (V + 1) x 4 u64, ONE at index 0 (pb.values layout), Montgomery unless canonical != 0 */
int zk_prove(zk_ctx *ctx, const uint64_t *witness, int canonical, zk_proof *out);
int zk_prove_timed(zk_ctx *ctx, const uint64_t *witness, int canonical, zk_proof *out, zk_timings *t);
/* sharded contexts: each rank computes its partial sums; after exchanging them (e.g. an RCCL
 * all-gather of the 640-byte structs), any rank folds them in rank order and finishes the proof */
int zk_prove_partial(zk_ctx *ctx, const uint64_t *witness, int canonical, zk_partials *out);
int zk_prove_partial_timed(zk_ctx *ctx, const uint64_t *witness, int canonical, zk_partials *out, zk_timings *t);
int zk_prove_combine(const zk_ctx *ctx, const zk_partials *parts, uint32_t count, zk_proof *out);
/* asynchronous form of zk_prove_partial: submit enqueues the whole proof on the context's streams and returns;
 * collect waits for it.  One proof in flight per context; several contexts keep the GPU full (ProverContext is
 * per concurrent prover in the reference too, hpp:279-291). */
int zk_prove_submit(zk_ctx *ctx, const uint64_t *witness, int canonical);
int zk_prove_collect(zk_ctx *ctx, zk_partials *out, zk_timings *t);
/* sharded provers, exchange on device buffers: after zk_prove_collect_device the four partial sums sit in a 640-byte device
 * buffer owned by the context (zk_ctx_partials_device; zk_partials layout) -- the send buffer of an RCCL all-gather;
 * zk_prove_combine_device folds `count` gathered 640-byte records (device pointer) in rank order like zk_prove_combine */
const void *zk_ctx_partials_device(const zk_ctx *ctx);
int zk_prove_collect_device(zk_ctx *ctx, zk_timings *t);
int zk_prove_combine_device(const zk_ctx *ctx, const void *d_parts, uint32_t count, zk_proof *out);
/* ---- several proofs of ONE circuit per launch sequence (the reference's nearest idea is the scratch reuse of
 * ProverContext, hpp:286-289): k <= zk_config.max_batch witnesses, contiguous (k x (V + 1) x 4 u64).  The k digit streams are
 * bucket-sorted together under the key (proof, bucket) and every kernel of the prover runs once for all of them, which is what
 * small circuits need: one proof alone is ~50 launches of latency-bound kernels.  Proof p is byte-identical to zk_prove of
 * witness p.  out: k records. */
int zk_prove_batch(zk_ctx *ctx, const uint64_t *witnesses, uint32_t k, int canonical, zk_proof *out);
int zk_prove_batch_submit(zk_ctx *ctx, const uint64_t *witnesses, uint32_t k, int canonical);
int zk_prove_batch_submit_resident(zk_ctx *ctx, const void *d_witnesses, uint32_t k, int canonical);
int zk_prove_batch_collect(zk_ctx *ctx, zk_partials *out, uint32_t k, zk_timings *t);
/* ---- sharded latency mode, SURVEY 8(e) option 2: the three transform chains of the witness map (row evaluations of A, B or C,
 * iFFT, cosetFFT) run on three different ranks instead of being replicated on all of them.
 *   zk_chain_submit         queue chain `which` (0 A, 1 B, 2 C) of this witness; its m results end up at zk_chain_device (A, B: evaluations on
 *                           the coset; C: coefficients divided by Z on the coset -- the witness map here needs six transforms, C never goes to the coset)
 *   zk_h_from_chains_submit queue (a b - c) / Z on the coset + icosetFFT from three chain buffers in THIS device's memory
 *                           (received from the other ranks); h ends up at zk_h_device (m elements)
 *   zk_chain_wait           wait for what was queued (check_degree: ZK_ERR_DEGREE unless h[m-1] = 0)
 *   zk_prove_submit_with_h  zk_prove_submit that takes this shard's coefficients of H -- h[lo .. hi), lo = (m-1) rank / count,
 *                           hi = (m-1) (rank+1) / count -- from a device buffer instead of computing H; collect as usual */
int zk_chain_submit(zk_ctx *ctx, const uint64_t *witness, int canonical, int which);
const void *zk_chain_device(const zk_ctx *ctx, int which);
int zk_h_from_chains_submit(zk_ctx *ctx, const void *dA, const void *dB, const void *dC);
const void *zk_h_device(const zk_ctx *ctx);
int zk_chain_wait(zk_ctx *ctx, int check_degree);
int zk_prove_submit_with_h(zk_ctx *ctx, const uint64_t *witness, int canonical, const void *d_h);
/* zk_prove_submit_with_h in two steps, so that every rank starts its witness sorts and A-, B-, L-query accumulations BEFORE H exists:
 *   zk_prove_submit_defer_h  queues everything that needs only the witness (upload, witness sorts, A-, B-, L-query);
 *                            zk_chain_submit (witness = NULL: the deferred proof's) / zk_h_from_chains_submit may follow on this context
 *   zk_prove_submit_h        queues the H-query from this shard's coefficients; collect as usual
 *   zk_prove_abort           drops a proof in flight (e.g. one that will not get its H part because rank 0 found the witness
 *                            unsatisfying): drains the context's streams, nothing is returned */
int zk_prove_submit_defer_h(zk_ctx *ctx, const uint64_t *witness, int canonical);
int zk_prove_submit_h(zk_ctx *ctx, const void *d_h);
int zk_prove_abort(zk_ctx *ctx);
/* zk_prove_submit for a witness that is already resident in the context's device memory (d_witness = device
 * pointer to (V + 1) x 32 bytes, e.g. written by a GPU witness generator); it must stay untouched until collected */
int zk_prove_submit_resident(zk_ctx *ctx, const void *d_witness, int canonical);
/* zk_prove_submit for a witness in PINNED host memory (SURVEY 8(d)'s metric: "witness already in pinned host memory"): the
 * asynchronous H2D copy reads the caller's buffer where it lies -- zk_prove_submit copies a pageable buffer into the context's
 * pinned staging buffer first, 33 MB of memcpy per proof at 2^20 --, so it must stay untouched until the proof is collected.
 * zk_host_alloc / zk_host_free: pinned host memory; zk_host_register / _unregister: pin a buffer the caller already owns
 * (pb.values of a protoboard that is proven repeatedly). */
int zk_prove_submit_pinned(zk_ctx *ctx, const uint64_t *witness, int canonical);
int zk_prove_batch_submit_pinned(zk_ctx *ctx, const uint64_t *witnesses, uint32_t k, int canonical);
int zk_host_alloc(size_t bytes, void **out);
int zk_host_free(void *p);
int zk_host_register(void *p, size_t bytes);
int zk_host_unregister(void *p);
/* double-buffered upload: zk_prove_stage copies the NEXT witness (k of them, k <= zk_config.max_batch) to the
 * device on a copy stream, also while a proof is in flight on this context; zk_prove_submit_staged starts that proof with no
 * upload on its critical path (collect as usual: zk_prove_collect / zk_prove_batch_collect).  One staged witness per context. */
int zk_prove_stage(zk_ctx *ctx, const uint64_t *witnesses, uint32_t k, int canonical);
int zk_prove_stage_pinned(zk_ctx *ctx, const uint64_t *witnesses, uint32_t k, int canonical);   /* pinned source: no staging memcpy; untouched until collected */
int zk_prove_submit_staged(zk_ctx *ctx);
/* info[3q .. 3q+2] = {window bits c, windows W, buckets 2^(c-1)} of query q = A, B, H, L; info[12..14] = the A-, B-, L-query
 * ride the shared witness sort; info[15] = domain size m */
int zk_ctx_info(const zk_ctx *ctx, uint32_t info[16]);
/* the window-multiple tables of the context's key shard: info = {bytes they take, bytes they would take with every window tabulated,
 * planes S, table rows of the B-query}.  S = 1 is the default layout (all W windows: 16 x the key).  When that does not fit the device --
 * the reference's domain goes up to 2^28, src/stubs.cpp:49-75 -- zk_ctx_create keeps every S-th window only (S = 2, 4, 8, 16) and proves
 * with S bucket planes instead of failing; the proof bytes are the same.  ZK_TABLE_BUDGET=<bytes> in the environment at context creation
 * caps the tables below the free device memory. */
int zk_ctx_table_info(const zk_ctx *ctx, uint64_t info[4]);

/* inputs: nIn Fr elements = witness[1..nIn] (Montgomery unless canonical); returns the JSON length
 * (excluding NUL) through *len; ZK_ERR_BUFFER if cap is too small (len still set) */
int zk_proof_to_json(const zk_proof *proof, const uint64_t *inputs, uint32_t nIn, int canonical,
                     char *buf, size_t cap, size_t *len);

/* ---- verifier (host code, as in the reference): r1cs_gg_ppzksnark_zok_verifier_strong_IC
 * (tcc:552-670) behind stub_verify (src/stubs.cpp:16-33).  *accepted = 1 iff the proof verifies. */
int zk_verify(const char *vk_json, const char *proof_json, int *accepted);
/* same symbol and signature as the reference's libethsnarks_verify (src/verify_dll.cpp:3-10) */
bool ethsnarks_verify(const char *vk_json, const char *proof_json);

/* ---- witness completion on the GPU (the reference fills pb.values on the host, gadget by gadget).  For a
 * constraint system in "solved order" -- every constraint reads known variables in A and B and introduces at most one new
 * variable, linearly, in C (the MiMC / Merkle gadgets, the synthetic chain) -- the system itself is the witness program.
 *   zk_wplan_create  compiles it; known[v] != 0 marks the variables the caller will supply (V + 1 flags; ONE is implied);
 *                    ZK_ERR_ARG with an explanatory message when the constraints are not in solved order
 *   zk_wplan_solve   completes k witnesses in place: d_w = device pointer to k x (V + 1) Fr elements (Montgomery), supplied
 *                    variables filled in; *violations = constraints that introduce nothing and do not hold (0 = all satisfied).
 *                    The buffer can go straight to zk_prove_batch_submit_resident: the witnesses never visit the host.
 * zk_dev_*: device-memory helpers for hosts without a HIP binding of their own. */
int zk_wplan_create(const zk_csr *A, const zk_csr *B, const zk_csr *C, uint32_t nC, uint32_t V, const uint8_t *known, int device, zk_wplan **out);
/* ... with HINTS for values the constraints only check (gadgets with non-deterministic advice, e.g. src/gadgets/field2bits_strict.cpp):
 *   ZK_WHINT_BITS      w[first + i] = bit i of the canonical value of w[src], i < count
 *   ZK_WHINT_INV       w[first] = 1 / w[src], 0 when w[src] = 0            (count = 1; src/gadgets/isnonzero.cpp: M)
 *   ZK_WHINT_NONZERO   w[first] = 1 when w[src] != 0, else 0               (count = 1; src/gadgets/isnonzero.cpp: Y)
 * A hint runs right before the first constraint that reads one of its variables (its source must be known by then). */
#define ZK_WHINT_BITS 1
#define ZK_WHINT_INV 2
#define ZK_WHINT_NONZERO 3
typedef struct { uint32_t kind, src, first, count; } zk_whint;
int zk_wplan_create_hinted(const zk_csr *A, const zk_csr *B, const zk_csr *C, uint32_t nC, uint32_t V, const uint8_t *known,
                           const zk_whint *hints, uint32_t n_hints, int device, zk_wplan **out);
int zk_wplan_solve(zk_wplan *plan, void *d_w, uint32_t k, uint32_t *violations);
void zk_wplan_free(zk_wplan *plan);
int zk_dev_alloc(size_t bytes, int device, void **out);
int zk_dev_free(void *p);
int zk_dev_upload(void *dst, const void *src, size_t bytes);
int zk_dev_download(void *dst, const void *src, size_t bytes);

/* ---- measurement aids (bench.py): kernel launches issued by this library so far; between zk_profile_begin() and
 * zk_profile_end() every launch is bracketed by a HIP event pair on its own stream -- the sum of the kernel durations
 * (overlapping kernels counted each), their number, and "name calls ms" lines per kernel come back */
uint64_t zk_launch_count(void);
int zk_profile_begin(void);
int zk_profile_end(float *kernel_ms_sum, uint32_t *launches, char *buf, size_t cap);
int zk_device_info(int device, uint32_t *compute_units, uint32_t *clock_mhz, char *name, size_t name_cap);
/* "domain:bus:device.function" of the HIP device (multi-GPU runs report it per rank: proof that the ranks sit on distinct GPUs) */
int zk_device_pci_bus_id(int device, char *buf, size_t cap);

/* ---- kernel-level entry points (parity tests / micro-benchmarks); host buffers in and out */
int zk_ntt(uint64_t *data, uint32_t logm, int inverse, int coset, int device);
int zk_witness_map(zk_ctx *ctx, const uint64_t *witness, int canonical, uint64_t *h_out /* (m+1) x 4 */);
int zk_msm_g1(const uint64_t *bases, const uint64_t *scalars, uint32_t n, uint32_t c, int device, uint64_t out_affine[8]);
int zk_msm_g2(const uint64_t *bases, const uint64_t *scalars, uint32_t n, uint32_t c, int device, uint64_t out_affine[16]);
/* host-only: n Fr elements between canonical and Montgomery form, in place (adapters whose field objects are opaque) */
int zk_fr_convert(uint64_t *io, uint32_t n, int to_montgomery);
int zk_field_mul(const uint64_t *a, const uint64_t *b, uint64_t *out, uint32_t n, int field /* 0 Fr, 1 Fq */, int device);

#ifdef __cplusplus
}
#endif
#endif
