/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE: C ABI of the CPU oracle (liboracle.so).
 *
 * CPU restatement of the reference proving path
 *   ethsnarks::prove                      src/stubs.cpp:42-47
 *   r1cs_gg_ppzksnark_zok_prover          src/r1cs_gg_ppzksnark_zok/r1cs_gg_ppzksnark_zok.tcc:451-550
 *   r1cs_gg_ppzksnark_zok_generator       ...tcc:277-449  (+ nozk conversion ...hpp:209-233)
 *   proof_to_json / vk2json               src/export.cpp:99-145
 *   operator<< / >> (pk_nozk)             ...tcc:108-143  (the .raw stream)
 * The arithmetic below them (libff / libfqfft / libsnark, CortexFoundation forks @ branch opt) is
 * ABSENT from /root/reference, so it is restated from the published algorithms.
 *
 * PARITY PINNING: the reference holds no (pk, witness) -> proof vector.  This oracle is pinned by
 *   (1) oracle/pyref.py's pairing verifier, itself validated on the reference's static triple
 *       (test/test_verify.py:10-12), accepting this oracle's proofs under this oracle's keys, and
 *   (2) byte-identical proof JSON against pyref.py's independent big-int prover (tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Conventions: field elements are 4 x u64 LE limbs, Montgomery form (R = 2^256) unless a name says
 * "canon".  Affine G1 = {x, y} (8 u64), affine G2 = {x.c0, x.c1, y.c0, y.c1} (16 u64); all-zero
 * coordinates encode the point at infinity.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t n_rows;
    const uint32_t *row_ptr;   /* n_rows + 1 */
    const uint32_t *col;       /* variable index, 0 = ONE */
    const uint64_t *coeff;     /* nnz x 4, Montgomery */
} orc_csr;

typedef struct {
    uint32_t nC, nIn, V;       /* constraints, public inputs, variables excluding ONE */
    orc_csr A, B, C;
} orc_r1cs;

/* canonical (non-Montgomery) affine coordinates; *_inf != 0 => point at infinity, coords (0,1) */
typedef struct {
    uint64_t a_x[4], a_y[4];
    uint64_t b_x_c0[4], b_x_c1[4], b_y_c0[4], b_y_c1[4];
    uint64_t c_x[4], c_y[4];
    uint32_t a_inf, b_inf, c_inf, _pad;
} orc_proof;

typedef struct orc_pk orc_pk;
typedef struct orc_vk orc_vk;

/* ---- field / domain primitives (for kernel-level parity tests) */
void orc_fr_to_mont(uint64_t *out, const uint64_t *in, size_t n);
void orc_fr_from_mont(uint64_t *out, const uint64_t *in, size_t n);
void orc_fq_to_mont(uint64_t *out, const uint64_t *in, size_t n);
void orc_fq_from_mont(uint64_t *out, const uint64_t *in, size_t n);
void orc_fr_mul(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t n);
void orc_fq_mul(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t n);
/* in-place size-2^logm transform over {omega^j}; inverse: includes 1/m; coset: shift g = 5
 * (libfqfft FFT / iFFT / cosetFFT / icosetFFT, Appendix A.2) */
void orc_ntt(uint64_t *a, uint32_t logm, int inverse, int coset);
uint32_t orc_domain_size(uint32_t nC, uint32_t nIn);               /* src/stubs.cpp:49-65 */
/* h[0..m] (m+1 elements, Montgomery) <- r1cs_to_qap_witness_map, call site tcc:461-468 */
int orc_witness_map(const orc_r1cs *cs, const uint64_t *witness, uint64_t *h_out);

/* ---- multi-exponentiation (tcc:488-530); scalars Montgomery Fr; result affine Montgomery */
void orc_msm_g1(const uint64_t *bases, const uint64_t *scalars, size_t n, unsigned c, uint64_t out_aff[8]);
void orc_msm_g2(const uint64_t *bases, const uint64_t *scalars, size_t n, unsigned c, uint64_t out_aff[16]);
void orc_msm_g1_naive(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_aff[8]);
void orc_msm_g2_naive(const uint64_t *bases, const uint64_t *scalars, size_t n, uint64_t out_aff[16]);
/* out[i] = scalars[i] * generator (fixed-base; G1 gen (1,2), G2 gen = standard alt_bn128 G2) */
void orc_batch_mul_g1(const uint64_t *scalars, size_t n, uint64_t *out_aff);
void orc_batch_mul_g2(const uint64_t *scalars, size_t n, uint64_t *out_aff);

/* ---- keys */
/* seeded generator: toxic waste (t, alpha, beta, gamma, delta) = 5 x (4 SplitMix64 draws mod r) */
int orc_keygen(const orc_r1cs *cs, uint64_t seed, orc_pk **pk_out, orc_vk **vk_out);
int orc_keygen_explicit(const orc_r1cs *cs, const uint64_t toxic_canon[20], orc_pk **pk_out, orc_vk **vk_out);
void orc_pk_free(orc_pk *);
void orc_vk_free(orc_vk *);
/* sizes[0..5] = A.domain, nA, B.domain, nB, nH, nL */
void orc_pk_sizes(const orc_pk *, uint32_t sizes[6]);
/* which: 0 alpha_g1 1 beta_g1 2 beta_g2 3 delta_g1 4 delta_g2 5 A.idx 6 A.val 7 B.idx 8 B.val 9 H 10 L */
const void *orc_pk_ptr(const orc_pk *, int which);
int orc_pk_from_parts(const uint64_t *alpha_g1, const uint64_t *beta_g1, const uint64_t *beta_g2,
                      const uint64_t *delta_g1, const uint64_t *delta_g2,
                      uint32_t a_domain, uint32_t nA, const uint32_t *a_idx, const uint64_t *a_val,
                      uint32_t b_domain, uint32_t nB, const uint32_t *b_idx, const uint64_t *b_val,
                      uint32_t nH, const uint64_t *H, uint32_t nL, const uint64_t *L, orc_pk **out);
int orc_pk_write_raw(const orc_pk *, const char *path);           /* ALT_BN128 / upstream libff layout */
int orc_pk_read_raw(const char *path, orc_pk **out);
size_t orc_vk_to_json(const orc_vk *, char *buf, size_t cap);     /* src/export.cpp:124-145 */

/* ---- prover */
/* phase_seconds[6]: H polynomial, A-query, B-query, H-query, L-query, final (names of tcc:460-542) */
int orc_prove(const orc_pk *, const orc_r1cs *cs, const uint64_t *witness, unsigned msm_c,
              orc_proof *out, double phase_seconds[6]);
/* the same proof in closed form from the toxic waste (t, alpha, beta, gamma, delta canonical, as orc_keygen_explicit takes
 * them): three scalar multiplications, no multi-exponentiation / transform / key involved (tcc:533-540 comments) */
int orc_proof_from_trapdoor(const orc_r1cs *cs, const uint64_t *witness, const uint64_t toxic_canon[20], orc_proof *out);
/* inputs: nIn Montgomery Fr elements (witness[1..nIn]); returns bytes needed (excluding NUL) */
size_t orc_proof_to_json(const orc_proof *, const uint64_t *inputs, uint32_t nIn, char *buf, size_t cap);
int orc_num_threads(void);
void orc_set_threads(int n);   /* OpenMP team size; callers set it to the container's CPU share */

#ifdef __cplusplus
}
#endif
#endif
