/* bpgpu -- C ABI of the MI355X (gfx950) backend for the MSM + IPA-folding hot path of
 * alex-ozdemir/bulletproofs (`ark-bp`).
 *
 * These are the entry points the reference's FFI shim for this path binds (see
 * INTEGRATION.md for the Rust `extern "C"` block and the call sites it replaces).
 * Plain pointers and sizes only; no exceptions cross the boundary; every function
 * returns 0 on success and a negative BPGPU_E_* code on failure, after which
 * bpgpu_last_error() describes it (the shim turns that into `panic!`, the
 * reference's only error convention on this path).
 *
 * Data formats (all little-endian 64-bit limbs, exactly ark-ff 0.3's in-memory
 * Fp256 = value * 2^256 mod p, canonical):
 *   scalar            : 4 x u64                       (32 B)
 *   jacobian point    : X[4] Y[4] Z[4], Z = 0 identity (96 B)   ark GroupProjective
 *   affine point      : x[4] y[4], (0,0) = identity    (64 B)
 *   result "xy_inf"   : x[4] y[4] inf, inf = 1 for the identity (72 B, x=y=0 then)
 * curve: 0 = Pallas, 1 = Vesta.
 *
 * Results are group elements in *normalised* (affine) form: the shim rebuilds
 * Projective { x, y, z: 1 } (or zero()), see SURVEY.md section 0.3.
 *
 * Threading: callable from any host thread; calls on one context (one per device: the
 * primary one of bpgpu_init, one per entry of bpgpu_sharded_init) are serialised by that
 * context's mutex (the reference never calls msm concurrently).  An IPA session runs on the
 * context it was opened on.
 * Ownership: the caller owns every buffer it passes; the library copies.
 */
#ifndef BPGPU_H
#define BPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPGPU_OK 0
#define BPGPU_E_ARG (-1)      /* bad argument (curve id, null pointer, size) */
#define BPGPU_E_HIP (-2)      /* a HIP runtime call failed */
#define BPGPU_E_NOINIT (-3)   /* bpgpu_init has not succeeded in this process */
#define BPGPU_E_STATE (-4)    /* session call in the wrong state */

#define BPGPU_PALLAS 0
#define BPGPU_VESTA 1

/* Size limits (the bucket sort keeps entry positions in 32 bits; 13 windows x 2^28 points < 2^32).  Larger
 * inputs are rejected with BPGPU_E_ARG before any device work; split them and add the partial results
 * (bpgpu_points_sum), or use bpgpu_msm_sharded. */
#define BPGPU_MSM_MAX_N ((size_t)1 << 28) /* points per bpgpu_msm* call (and per shard of a sharded MSM) */
#define BPGPU_IPA_MAX_N ((size_t)1 << 26) /* elements per vector of an IPA session */

/* Create (or reuse) the primary context on HIP device `device`: the one every single-device entry point
 * below runs on.  Idempotent.  Replaces nothing in the reference (it has no device).
 * bpgpu_shutdown releases every context; it is refused (nothing is released, bpgpu_last_error says why)
 * while an IPA session is still open, because the session's device memory belongs to its context. */
int bpgpu_init(int device);
void bpgpu_shutdown(void);
const char* bpgpu_last_error(void);
/* The three tuning knobs below are PROCESS-WIDE (every context, every thread): results never depend on
 * them, timings do.  MSM window width override (0 = automatic).  */
int bpgpu_set_window_bits(int c);
/* IPA sessions whose vectors are at most this long defer the generator fold ("deferred
 * folding"; default 2^14, 0 = always fold).  Tuning only: results are identical either way. */
int bpgpu_set_freeze_len(size_t len);
/* IPA sessions fold the generators of their long rounds two rounds at a time ("pair
 * folding"; default on).  Tuning only: results are identical either way. */
int bpgpu_set_pair_fold(int enable);
/* IPA sessions build a fixed-base table (the multiples 2^(15 w) of the generators as they are when folding stops)
 * for their short rounds when those generators number at least min_len ("fixed-base table"): 0 = never,
 * 1 = the default (2^12: twelve rounds left at least), otherwise the threshold itself.  Tuning only: results are
 * identical either way. */
int bpgpu_set_ipa_table(size_t min_len);
/* Back to the shipped defaults of all four knobs (automatic window, freeze length 2^14, pair folding on, table
 * from 2^12 frozen generators), and the current values: out[0] window bits (0 = automatic), out[1] freeze
 * length, out[2] pair folding (0/1), out[3] table threshold.  No reference counterpart (tuning only). */
void bpgpu_reset_tuning(void);
int bpgpu_get_tuning(size_t out[4]);

/* ---- variable-base MSM ------------------------------------------------------
 * Replaces `pippenger_msm` (= `crate::util::msm`), reference src/util/msm.rs:88-193
 * (entry re-exported at src/util/mod.rs:15).  The shim zips the two iterators into
 * contiguous buffers first (the reference's own `zip ... unzip`, msm.rs:94-97), so
 * `n` is the zipped length.  Accepts Z != 1, identity points, scalars 0 and 1, n = 0
 * (-> identity). */
int bpgpu_msm(int curve, const uint64_t* points_xyz, const uint64_t* scalars, size_t n,
              uint64_t out_xy_inf[9]);
/* Same with 64-byte affine inputs (known-affine generators). */
int bpgpu_msm_affine(int curve, const uint64_t* points_xy, const uint64_t* scalars, size_t n,
                     uint64_t out_xy_inf[9]);

/* ---- inner product ----------------------------------------------------------
 * Replaces `ip`, reference src/util/mod.rs:17-23 (caller asserts equal length). */
int bpgpu_ip(int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]);

/* ---- scalar-vector fold -----------------------------------------------------
 * out[i] = s_lo * v[i] + s_hi * v[i + n_out], i < n_out.  Replaces the map/collect at
 * reference src/proofs/bp_iter.rs:88-95 == src/reductions/bp_2ary_step.rs:51-58
 * (a: (s_lo, s_hi) = (x, x^-1); b: (x^-1, x)).  `out` may alias `v`. */
int bpgpu_fold_scalars(int curve, const uint64_t* v, size_t n_out, const uint64_t s_lo[4],
                       const uint64_t s_hi[4], uint64_t* out);

/* ---- generator fold ---------------------------------------------------------
 * out[i] = g[i] * s_lo + g[i + n_out] * s_hi.  Replaces reference
 * src/proofs/bp_iter.rs:97-104 == src/reductions/bp_2ary_step.rs:60-67,106-113 ==
 * src/relations/ipa.rs:54-63 (A: (x^-1, x); B: (x, x^-1)).  Inputs are jacobian
 * (96 B); outputs are jacobian with Z = 1 (identity: (0, 1, 0)). */
int bpgpu_fold_points(int curve, const uint64_t* g_xyz, size_t n_out, const uint64_t s_lo[4],
                      const uint64_t s_hi[4], uint64_t* out_xyz);

/* ---- device-resident variants ------------------------------------------------
 * For callers that keep generators / witnesses in HBM (the fixed public parameters
 * of the IPA).  Device pointers are plain `void*` obtained from bpgpu_dev_alloc (or
 * any hipMalloc'd memory on the context's device).  Device point format ("dev
 * affine", 64 B): x[8] y[8] u32 words, canonical value * 2^261 mod p, (0,0) =
 * identity -- produced only by bpgpu_points_to_device.  Device scalars are in ark
 * form (32 B), i.e. a plain upload of the host array. */
int bpgpu_dev_alloc(size_t bytes, void** dptr);
int bpgpu_dev_free(void* dptr);
int bpgpu_dev_upload(void* dptr, const void* host, size_t bytes);
int bpgpu_dev_download(void* host, const void* dptr, size_t bytes);
/* fmt: 0 = jacobian 96 B, 1 = affine 64 B.  d_out receives n * 64 bytes. */
int bpgpu_points_to_device(int curve, const uint64_t* points, size_t n, int fmt, void* d_out);
/* d_points (n * 64 B dev affine) -> host jacobian Z=1 / (0,1,0) */
int bpgpu_points_from_device(int curve, const void* d_points, size_t n, uint64_t* out_xyz);
int bpgpu_msm_device(int curve, const void* d_points, const void* d_scalars, size_t n,
                     uint64_t out_xy_inf[9]);
int bpgpu_ip_device(int curve, const void* d_a, const void* d_b, size_t n, uint64_t out[4]);
/* bpgpu_ip_device for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, code that must not disturb the allocator).  NO COUNTERPART IN THE REFERENCE.  Same inputs; the
 * result (one ark-form scalar, 32 B) is written to caller-owned DEVICE memory d_out; the block partial sums
 * live in a fixed 32 KiB file-scope device array instead of a grow-only workspace, and the call performs no
 * hipMalloc / hipFree / allocating library routine.  It launches on the library's own stream and returns after
 * that stream has drained (so it is not itself capturable into a HIP graph).  n <= BPGPU_IP_NOALLOC_MAX_N,
 * larger n is BPGPU_E_ARG.  The array is per process and per GPU: calls through one context are serialised by
 * the library, but two contexts of one process on the same GPU (bpgpu_sharded_init with a repeated device, or a
 * shard context beside the primary one) share it, and such calls must not overlap. */
#define BPGPU_IP_NOALLOC_MAX_N ((size_t)0xffffffffu)
int bpgpu_ip_device_noalloc(int curve, const void* d_a, const void* d_b, size_t n, void* d_out);
int bpgpu_fold_scalars_device(int curve, const void* d_v, size_t n_out, const uint64_t s_lo[4],
                              const uint64_t s_hi[4], void* d_out);
int bpgpu_fold_points_device(int curve, const void* d_g, size_t n_out, const uint64_t s_lo[4],
                             const uint64_t s_hi[4], void* d_out);

/* ---- vector utilities of the k-ary protocols ("next" rows: bp_rec_kary, ipa_to_bp_unroll) ------
 * Generalised fold: out[i] = sum_{j<k} y[j] * v[j*ck + i], i < ck.  Replaces
 *   sum_vecs(chunks.iter().zip(&pows).map(|(c, y)| scale_vec(y, c)), ck_size)
 * (scale_vec src/util/mod.rs:25-38, sum_vecs :40-55) at reference src/proofs/bp_rec_kary.rs:100-112,
 * :171-178 and src/reductions/ipa_to_bp_unroll.rs:248-260, :352-359; the caller has already padded
 * the vector to k*ck (zero_pad_to_multiple / zero_intersperse_to_multiple).  k = 1 is scale_vec,
 * y = (1, ..., 1) is sum_vecs, k = 0 gives zeros / identities (sum_vecs of nothing).  y: k scalars,
 * host memory.  Scalars: k <= 1024; `out` may alias `v`.  Points: jacobian in, jacobian Z = 1 out;
 * with three chunks or more the multipliers share one doubling chain, seven at a time (chunk 0 is simply
 * added when y[0] = 1, as in the reference's powers(x, 0..k)); bpgpu_last_fold_ops reports the group
 * operations of the whole call. */
int bpgpu_fold_scalars_kary(int curve, const uint64_t* v, size_t ck, size_t k, const uint64_t* y,
                            uint64_t* out);
int bpgpu_fold_points_kary(int curve, const uint64_t* g_xyz, size_t ck, size_t k, const uint64_t* y,
                           uint64_t* out_xyz);
/* device-resident forms (dev affine points, ark scalars); d_out must not overlap d_g for points */
int bpgpu_fold_scalars_kary_device(int curve, const void* d_v, size_t ck, size_t k, const uint64_t* y,
                                   void* d_out);
int bpgpu_fold_points_kary_device(int curve, const void* d_g, size_t ck, size_t k, const uint64_t* y,
                                  void* d_out);
/* Cross terms of one k-ary round.  Replaces the closure `x(i, j)` and the two sums over it at reference
 * src/reductions/ipa_to_bp_unroll.rs:189-215 == src/proofs/bp_rec_kary.rs:52-76:
 *   X[i, j] = msm(a_gen[j] | b_gen[i] | Q,  a[i] | b[j] | ip(a[i], b[j]))    (chunks of length ck of the padded vectors)
 *   pos[t]  = sum_{j < k-t-1} X[t+j+1, j],    neg[t] = sum_{j < k-t-1} X[j, t+j+1],    t < k - 1.
 * a_gens_xyz, b_gens_xyz: k * ck jacobian points, a, b: k * ck scalars (the caller has padded them:
 * zero_pad_to_multiple / zero_intersperse_to_multiple); pos_xy_inf, neg_xy_inf receive k - 1 results (72 B each).
 * The k (k - 1) MSMs are merged into 2 (k - 1) over contiguous operand ranges and run as one batch; the group elements
 * are the reference's.  k < 2: nothing is written. */
int bpgpu_kary_cross_terms(int curve, const uint64_t* a_gens_xyz, const uint64_t* b_gens_xyz,
                           const uint64_t* ip_gen_xyz, const uint64_t* a, const uint64_t* b, size_t ck, size_t k,
                           uint64_t* pos_xy_inf, uint64_t* neg_xy_inf);
/* out[i] = a[i] * b[i].  Replaces `hadamard`, reference src/util/mod.rs:80-85 (caller asserts
 * equal length). */
int bpgpu_hadamard(int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* out[i] = bases[i] * scalars[i] (every element its own multiplier).  Replaces `hadamard_gp`,
 * reference src/util/mod.rs:70-78 (used at src/reductions/com_r1cs_to_ipa.rs:110). */
int bpgpu_hadamard_points(int curve, const uint64_t* bases_xyz, const uint64_t* scalars, size_t n,
                          uint64_t* out_xyz);
/* out[i] = f^(start + i), i < n.  Replaces `powers(f, start..start+n)`, reference
 * src/util/mod.rs:57-64 (`neg_powers` = powers of the inverse, :66-68); f^0 = 1 for every f. */
int bpgpu_powers(int curve, const uint64_t f[4], uint64_t start, size_t n, uint64_t* out);

/* ---- sparse matrix - vector products of the Com-R1CS compiler ("next" row f4) --------------------
 * The reference's Matrix<F> is a list of rows of (value, column) pairs; the shim flattens it into
 * compressed rows: row r holds entries row_ptr[r] .. row_ptr[r+1] of col_idx / vals (row_ptr has
 * n_rows + 1 elements, row_ptr[0] = 0; indices are `usize`, hence u64).
 * bpgpu_mat_vec_mult: out[r] = sum_e vals[e] * vec[col_idx[e]], r < n_rows.  Replaces `mat_vec_mult`,
 *   reference src/relations/com_r1cs.rs:43-57 (called at src/reductions/com_r1cs_to_ipa.rs:63-64);
 *   a column index >= vec_len is an error (the reference asserts, :51).
 * bpgpu_vec_mat_mult: out[c] = sum over entries e of row r with col_idx[e] = c of vals[e] * vec[r],
 *   c < output_size.  Replaces `vec_mat_mult`, com_r1cs.rs:67-83 (com_r1cs_to_ipa.rs:93-99); errors where
 *   the reference asserts: vec_len != n_rows (:74), a column >= output_size (:72). */
int bpgpu_mat_vec_mult(int curve, const uint64_t* row_ptr, const uint64_t* col_idx, const uint64_t* vals,
                       size_t n_rows, const uint64_t* vec, size_t vec_len, uint64_t* out);
int bpgpu_vec_mat_mult(int curve, const uint64_t* vec, size_t vec_len, const uint64_t* row_ptr,
                       const uint64_t* col_idx, const uint64_t* vals, size_t n_rows, size_t output_size,
                       uint64_t* out);

/* ---- seed-faithful point sampling ("next" row f4) ---------------------------------------------
 * The data-parallel half of `G::rand` (ark-ec 0.3 GroupProjective::rand for y^2 = x^3 + 5) as the
 * reference's input recipes use it: examples/msm_bench.rs:41, src/relations/ipa.rs:28-36.  The caller
 * walks its RNG stream -- per attempt x = Fq::rand (rejection), then one bool; the draws do not depend
 * on whether the attempt succeeds -- and passes every candidate: xs[i] (ark form), greatest[i] (0/1).
 * ok[i] = 1 iff x^3 + 5 is a square, and then ys[i] (ark form) is the root ark selects:
 * (y < -y) XOR greatest ? y : -y, compared as canonical integers.  ys[i] = 0 where ok[i] = 0. */
int bpgpu_curve_lift_x(int curve, const uint64_t* xs, const uint8_t* greatest, size_t m, uint64_t* ys,
                       uint8_t* ok);

/* ---- device-resident IPA prover session -------------------------------------
 * Lets `Bp::prove` (reference src/proofs/bp_iter.rs:49-128) keep a, b, A, B in HBM
 * across its log2(N) rounds; Fiat-Shamir stays with the caller between
 * `round_lr` and `round_fold`.  Inputs are the *unpadded* vectors; the session pads
 * to a power of two with identity points / zero scalars exactly as
 * zero_pad_to_two_power does (src/util/mod.rs:137-144, bp_iter.rs:59-62).
 *
 *   begin -> { round_lr -> round_fold } x log2(N') -> finish -> free
 *
 * round_lr computes (bp_iter.rs:68-81)
 *   L = msm(A[n..] | B[..n] | Q,  a[..n] | b[n..] | <a[..n], b[n..]>)
 *   R = msm(A[..n] | B[n..] | Q,  a[n..] | b[..n] | <a[n..], b[..n]>)
 * round_fold(x, x_inv) applies bp_iter.rs:88-104.  export materialises the current
 * (a_gens, b_gens, a, b) for `Bp2aryStep::prove`, which must return the folded
 * instance (src/reductions/bp_2ary_step.rs:72-86). */
typedef struct bpgpu_ipa bpgpu_ipa;
int bpgpu_ipa_begin(int curve, const uint64_t* a_gens_xyz, const uint64_t* b_gens_xyz,
                    const uint64_t* ip_gen_xyz, const uint64_t* a, const uint64_t* b, size_t n,
                    bpgpu_ipa** out);
/* Same session, generators already in HBM: the IPA generators are fixed public parameters
 * (src/relations/ipa.rs:28-36 samples them once per relation size), so a prover that makes many
 * proofs uploads them once with bpgpu_points_to_device and pays only the witness upload per
 * proof.  d_a_gens, d_b_gens: n device affine points each; d_ip_gen: one.  The session works on
 * its own copy; the caller's arrays are not modified. */
int bpgpu_ipa_begin_device(int curve, const void* d_a_gens, const void* d_b_gens, const void* d_ip_gen,
                           const uint64_t* a, const uint64_t* b, size_t n, bpgpu_ipa** out);
/* current (padded) vector length; 1 when the loop is finished */
size_t bpgpu_ipa_len(const bpgpu_ipa* s);
int bpgpu_ipa_round_lr(bpgpu_ipa* s, uint64_t l_xy_inf[9], uint64_t r_xy_inf[9]);
/* round_fold copies x and x_inv before it returns.  Once the session's generators are frozen (short vectors) it
 * returns without waiting for its device work: a device error of that work is reported by the next call on the
 * session that waits (round_lr, finish, export), which then marks the session failed like any other error. */
int bpgpu_ipa_round_fold(bpgpu_ipa* s, const uint64_t x[4], const uint64_t x_inv[4]);
int bpgpu_ipa_finish(bpgpu_ipa* s, uint64_t final_a[4], uint64_t final_b[4]);
int bpgpu_ipa_export(bpgpu_ipa* s, uint64_t* a_gens_xyz, uint64_t* b_gens_xyz, uint64_t* a,
                     uint64_t* b);
void bpgpu_ipa_free(bpgpu_ipa* s);

/* ---- verifier MSM ("next" row f1) ----------------------------------------------------
 * The right-hand side of Bp::verify, reference src/proofs/bp_iter.rs:130-178, as one call:
 *   msm( A | B | Q | L_0 R_0 L_1 R_1 ...,
 *        final_a * s | final_b * s_inv | final_a * final_b | -x_0^2, -x_0^-2, -x_1^2, ... )
 * with s = scalars_from_challenges(challenges), s_inv = scalars_from_challenges(inverses) (:25-42)
 * expanded on the device.  The caller replays Fiat-Shamir to obtain `challenges` (k of them, :146-153)
 * and compares the returned point with instance.result (:176).  n: unpadded generator count; the
 * generators are zero-padded to 2^k with identity points as :137-138 does; k must be log2 of that
 * padded length.  A zero challenge is an error (the reference's inverse().unwrap() panics). */
int bpgpu_ipa_verify_msm(int curve, const uint64_t* a_gens_xyz, const uint64_t* b_gens_xyz,
                         const uint64_t* ip_gen_xyz, size_t n, const uint64_t* ls_xyz,
                         const uint64_t* rs_xyz, size_t k, const uint64_t* challenges,
                         const uint64_t final_a[4], const uint64_t final_b[4], uint64_t out_xy_inf[9]);
/* same with the generators already in HBM (dev affine, bpgpu_points_to_device) */
int bpgpu_ipa_verify_msm_device(int curve, const void* d_a_gens, const void* d_b_gens, const void* d_ip_gen,
                                size_t n, const uint64_t* ls_xyz, const uint64_t* rs_xyz, size_t k,
                                const uint64_t* challenges, const uint64_t final_a[4],
                                const uint64_t final_b[4], uint64_t out_xy_inf[9]);

/* ---- sharded (multi-GPU) MSM -------------------------------------------------
 * `pippenger_msm` (reference src/util/msm.rs:88-92; call sites src/proofs/bp_iter.rs:68-81) runs in ONE
 * process.  For that process to use several GPUs the library keeps one context per entry of `devices`
 * (HIP device numbers; NULL = 0 .. n_ctx-1; a number may repeat -- several contexts on one GPU -- which is
 * how the path is rehearsed on a one-GPU machine).  bpgpu_msm_sharded cuts the zipped pairs into contiguous
 * ranges [g n / P, (g+1) n / P), runs the single-GPU pipeline on every range at once (one persistent host
 * thread per context) and adds the P partial results on the host: each context's result is produced in host
 * memory (72 B), so no device-side collective is involved in the one-process form.  Same argument meaning,
 * edge cases and result as bpgpu_msm.  Independent of bpgpu_init (the single-device entry points keep using
 * the primary context). */
int bpgpu_sharded_init(int n_ctx, const int* devices);
int bpgpu_sharded_count(void);
int bpgpu_msm_sharded(int curve, const uint64_t* points_xyz, const uint64_t* scalars, size_t n,
                      uint64_t out_xy_inf[9]);
int bpgpu_msm_sharded_affine(int curve, const uint64_t* points_xy, const uint64_t* scalars, size_t n,
                             uint64_t out_xy_inf[9]);
/* Generators resident across calls (the IPA generators are fixed public parameters, src/relations/ipa.rs:28-36):
 * load once -- every context keeps its range in HBM as dev affine points -- then one call per scalar vector
 * (n scalars in host memory; each context uploads its range).  fmt: 0 = jacobian 96 B, 1 = affine 64 B. */
typedef struct bpgpu_sharded_gens bpgpu_sharded_gens;
int bpgpu_sharded_gens_load(int curve, const uint64_t* points, size_t n, int fmt, bpgpu_sharded_gens** out);
int bpgpu_sharded_gens_msm(bpgpu_sharded_gens* g, const uint64_t* scalars, uint64_t out_xy_inf[9]);
void bpgpu_sharded_gens_free(bpgpu_sharded_gens* g);
/* One process per GPU instead (bench.py, bulletproofs_amd/sharded.py): each rank runs bpgpu_msm* on its
 * range, the per-rank results (72 B each) are gathered (RCCL all_gather, done by the host language) and
 * summed here.  Host-only; no reference counterpart (the reference is single-process). */
int bpgpu_points_sum(int curve, const uint64_t* xy_inf, size_t count, uint64_t out_xy_inf[9]);

/* ---- instrumentation ----------------------------------------------------------
 * Duration in milliseconds of the GPU phases of the last bpgpu_msm* call on the primary
 * context (whichever thread made it), measured with HIP events on the library's stream:
 * [0] digits + coarse histogram, [1] bin scan + scatter to bins, [2] sort inside bins,
 * [3] bucket accumulation (k_accum + k_heavy_merge), [4] unused, [5] bucket reduction,
 * [6] host tail (wall), [7] total GPU.  The MSMs of an IPA session's short rounds (fixed-base table mode) record no
 * phase events: after one of them the GPU entries read 0. */
int bpgpu_last_msm_timings(float out_ms[8]);
/* Number of (point, window) pairs with a non-zero digit in the last bpgpu_msm* call (a batch counts both
 * of its MSMs): the additions the bucket accumulation performed, which bench.py prices the kernel with. */
int bpgpu_last_msm_entries(uint64_t* entries);
/* Duration (HIP events on the library's stream) of the kernels of the last inner-product, scalar-fold
 * or generator-fold launch: what bench.py prices against the HBM roofline for the scalar kernels. */
int bpgpu_last_op_ms(float* ms);
/* Group operations of the last generator-fold launch on the primary context, summed over its lanes:
 * [0] outputs, [1] doublings, [2] mixed additions (non-zero digits of the recoded shared multipliers, plus
 * the final P + ...), [3] kernel (1 k_fold_monic, 2 k_fold_joint4, 3 k_fold_points, 4 k_fold_multi: then summed over the launches
 * of the call): what bench.py prices
 * the fold kernels with. */
int bpgpu_last_fold_ops(uint64_t out[4]);
/* Sustained v_mad_u64_u32 issue rate (lane-ops/s) measured by a register-only loop:
 * the integer roofline bench.py prices the MSM against. */
int bpgpu_measure_mad_peak(double* mads_per_sec);

#ifdef __cplusplus
}
#endif
#endif /* BPGPU_H */
