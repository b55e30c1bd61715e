// Device kernels of the MSM + IPA-folding hot path (gfx950).  Host entry points are in
// bpgpu.hip; arithmetic is in field.cuh / curve.cuh.
//
// MSM pipeline (reference algorithm being replaced: pippenger_msm, src/util/msm.rs:88-193):
//   k_part_count / k_bin_scan / k_part_scatter / k_bin_count / k_bin_offsets / k_bin_place
//                    scalar -> canonical -> signed c-bit digits; two-level counting sort of
//                    the (point, window) pairs by bucket (LDS histograms, see below)
//   k_accum          persistent waves: one lane per bucket (one wave per piece of a heavy bucket),
//                    mixed additions into an XYZZ accumulator; k_heavy_merge sums the pieces
//   k_reduce1        one lane per run of S buckets: running sums (msm.rs:167-171) -> (T_r, S_r)
//   k_reduce2        one workgroup per (window, component): tree-sums of T_r and of S_r by index bit
// The Horner combine (msm.rs:177-190) over the windows' components is a few hundred dependent
// group operations and runs on the host (host_field.h).
#pragma once
#include <hip/hip_runtime.h>

#include "curve.cuh"

namespace bpgpu {

// One MSM operand is the concatenation of up to 3 ranges of a device point array and a
// device scalar array (the prover's L/R operands, bp_iter.rs:68-81, are such
// concatenations of halves of A, B, a, b plus the single pair (Q, <a,b>)).
struct MsmSeg {
  uint32_t pt_off, sc_off, count;
};
constexpr int MSM_MAX_BATCH = 32;
struct MsmPlan {
  int M;       // number of independent MSMs in this batch (the prover's L, R pair: 2; a k-ary round's cross terms: up to MSM_MAX_BATCH)
  int W;       // windows per MSM
  int c;       // window width in bits; digits are signed, |d| <= 2^(c-1)
  uint32_t n;  // slots per MSM: the sum of its segment counts, or the largest such sum when the MSMs of a batch differ
               // in length (the slots of a shorter one past its own end are padding: zero digits, no entries)
  uint32_t n_eff;  // slots whose scalar can be non-zero, if the caller knows (0 = n): sizes the windows
  // Fixed-base table mode (0 = off): the point array holds, for every point index p < table_stride, its multiples
  // 2^(c w) P at index w * table_stride + p (w < W).  Window w of a slot then takes the w-th multiple, every
  // window's digits fall into ONE bucket set per MSM (W times fewer buckets to reduce) and the window sum needs
  // no doublings.  c is then given by the caller (the width the table was built for).
  uint32_t table_stride;
  MsmSeg seg[MSM_MAX_BATCH][3];
};

__device__ __forceinline__ uint32_t extract_bits(const uint32_t (&w)[8], int pos, int c) {
  // bits [pos, pos+c) of the 256-bit little-endian value, c <= 24
  int wi = pos >> 5, sh = pos & 31;
  if (wi >= 8) return 0;
  uint64_t v = w[wi];
  if (wi + 1 < 8) v |= (uint64_t)w[wi + 1] << 32;
  return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// ---------------------------------------------------------------------------
// Bucket sort of the (point, window) pairs: a two-level counting sort whose only global
// atomics are one per (workgroup, coarse bin).
//   bucket id  = (m*W + j) << (c-1) | (|digit| - 1)              (global, over all windows)
//   coarse bin = bucket id >> fine_bits,  fine = bucket id & (2^fine_bits - 1)
//   k_part_count    digits (kept, window-major) -> per-workgroup LDS histogram of coarse bins -> bin_count
//   k_bin_scan      exclusive scan of the bin counts (one small workgroup)
//   k_part_scatter  digit words -> (index|sign, fine) entries, grouped by bin in LDS, appended to their bin
//   k_bin_count / k_bin_offsets / k_bin_place   counting sort of each bin by fine bucket -> sorted[], offs[]
// ---------------------------------------------------------------------------
constexpr uint32_t HEAVY_T_MIN = 128;   // a single lane never walks a list longer than max(this, 4 x the mean)
constexpr uint32_t HEAVY_PIECE_MAX = 1024;  // most entries per wave-piece of a heavy bucket (HeavyLists::piece)
struct HeavyLists {
  uint32_t threshold;  // lists longer than this are heavy
  uint32_t piece;      // entries per wave-piece: 64 lanes x about the entries an ordinary lane walks (256 .. HEAVY_PIECE_MAX)
  uint32_t* counters;  // [0] heavy buckets, [1] pieces, [2] piece work counter (all zeroed per MSM)
  uint32_t* bucket;    // per heavy bucket: bucket id
  uint32_t* piece0;    //                   first piece
  uint32_t* npieces;   //                   number of pieces
  uint32_t* piece_lo;  // per piece: [lo, hi) in sorted[]
  uint32_t* piece_hi;
  uint32_t* psum;      // per piece: XYZZ partial sum (32 words)
};
struct SortPlan {
  int fine_bits;       // buckets per bin = 2^fine_bits (<= 1024)
  uint32_t nbins;      // M * W << (c - 1 - fine_bits)
  uint32_t tiles;      // number of 1024-slot tiles = ceil(M*n / 1024)
};
constexpr int BIN_PAD = 16;  // bin counters live 64 B apart (one per L2 line)

// Exclusive prefix sum of v over the 1024 threads of a workgroup (total = sum over all of them): wave
// scans by lane shuffles, then the 16 wave totals through LDS -- two barriers where a 10-step
// Hillis-Steele scan in LDS takes twenty, and these sort kernels do only 8-16 entries per thread
// per scan.  wsum: 16 words of LDS, free again after the call's second barrier.
__device__ __forceinline__ uint32_t wg_excl_scan_1024(uint32_t v, uint32_t* wsum, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= (uint32_t)d) inc += t;
  }
  __syncthreads();  // a previous call's readers are done with wsum
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; w++) {
    uint32_t sw = wsum[w];
    tot += sw;
    if (w < wave) base += sw;
  }
  total = tot;
  return base + inc - v;
}

// slot -> (point index, scalar index); returns 0 past the end of the batch, 1 for a pair, 2 for a padding slot (past the
// end of an MSM shorter than plan.n)
__device__ __forceinline__ int slot_lookup(const MsmPlan& plan, uint64_t gid, int& m, uint32_t& pt_idx,
                                           uint32_t& sc_idx) {
  if (gid >= (uint64_t)plan.M * plan.n) return 0;
  m = (int)(gid / plan.n);
  uint32_t t = (uint32_t)(gid % plan.n);
  int s = 0;
  while (s < 2 && t >= plan.seg[m][s].count) {
    t -= plan.seg[m][s].count;
    s++;
  }
  if (t >= plan.seg[m][s].count) return 2;
  pt_idx = plan.seg[m][s].pt_off + t;
  sc_idx = plan.seg[m][s].sc_off + t;
  return 1;
}

// digit word of a (slot, window): |digit| | sign << 31; 0 = zero digit, no entry
// SF = scalar field id.  Canonical scalar (into_repr, msm.rs:96) -> signed digits, stored window-major
// (dig[(m*W + j)*n + t]) for the scatter pass, and counted into the coarse-bin histogram.
template <int SF>
__global__ void __launch_bounds__(1024) k_part_count(MsmPlan plan, SortPlan sp, const uint32_t* __restrict__ scalars,
                                                     uint32_t* __restrict__ dig, uint32_t* __restrict__ bin_count) {
  using F = Fp<SF>;
  extern __shared__ uint32_t lds[];
  uint32_t* cnt = lds;
  for (uint32_t b = threadIdx.x; b < sp.nbins; b += 1024) cnt[b] = 0;
  __syncthreads();
  const int c = plan.c, W = plan.W;
  const uint32_t half = 1u << (c - 1);
  for (uint32_t tile = blockIdx.x; tile < sp.tiles; tile += gridDim.x) {
    uint64_t gid = (uint64_t)tile * 1024 + threadIdx.x;
    int m;
    uint32_t pt_idx, sc_idx;
    const int kind = slot_lookup(plan, gid, m, pt_idx, sc_idx);
    if (kind == 0) continue;
    const uint32_t t = (uint32_t)(gid - (uint64_t)m * plan.n);
    if (kind == 2) {  // padding: zero digits in every window
      uint32_t* dz = dig + (uint64_t)m * W * plan.n + t;
      for (int j = 0; j < W; j++) dz[(uint64_t)j * plan.n] = 0u;
      continue;
    }
    const uint4* sp4 = reinterpret_cast<const uint4*>(scalars + 8ull * sc_idx);
    uint4 lo = sp4[0], hi = sp4[1];
    uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    // words < 2^256 times the plain constant 32: the reduced value lies in (-p, 1], one conditional addition away
    Fe v = F::mul(F::unpack(w), F::from_const(F::C::C256TOCANON));
    F::pack(F::canon_small(v), w);
    uint32_t carry = 0;
    // bucket set of window j: m W + j, or m alone in table mode (all windows share one set)
    const uint32_t wb0 = plan.table_stride ? (uint32_t)m : (uint32_t)(m * W), wbs = plan.table_stride ? 0u : 1u;
    uint32_t* dp = dig + (uint64_t)m * W * plan.n + t;
    for (int j = 0; j < W; j++) {
      uint32_t bits = extract_bits(w, j * c, c) + carry;
      uint32_t neg = bits > half ? 1u : 0u;
      uint32_t mag = neg ? (1u << c) - bits : bits;
      carry = neg;
      dp[(uint64_t)j * plan.n] = mag | (neg << 31);
      if (mag) atomicAdd(&cnt[(((wb0 + wbs * (uint32_t)j) << (c - 1)) + (mag - 1)) >> sp.fine_bits], 1u);
    }
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < sp.nbins; b += 1024) {
    uint32_t v = cnt[b];
    if (v) atomicAdd(&bin_count[(uint64_t)b * BIN_PAD], v);
  }
}

// bin_start[0..nbins] = exclusive scan of bin_count, tile_start[0..nbins] = exclusive scan of the
// bins' tile counts (a tile = up to BIN_TILE entries of one bin); zeroes bin_cursor.  One workgroup.
constexpr uint32_t BIN_TILE = 16384;
__global__ void __launch_bounds__(1024) k_bin_scan(const uint32_t* __restrict__ bin_count, uint32_t* __restrict__ bin_start,
                                                   uint32_t* __restrict__ tile_start, uint32_t* __restrict__ bin_cursor,
                                                   uint32_t nbins, uint32_t* __restrict__ total_out) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t tpart[1024];
  uint32_t tid = threadIdx.x;
  uint32_t per = (nbins + 1023) / 1024;
  uint32_t lo = min(tid * per, nbins), hi = min(lo + per, nbins);
  uint32_t s = 0, ts = 0;
  for (uint32_t i = lo; i < hi; i++) {
    uint32_t v = bin_count[(uint64_t)i * BIN_PAD];
    s += v;
    ts += (v + BIN_TILE - 1) / BIN_TILE;
  }
  part[tid] = s;
  tpart[tid] = ts;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = tid >= d ? part[tid - d] : 0;
    uint32_t tv = tid >= d ? tpart[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    tpart[tid] += tv;
    __syncthreads();
  }
  uint32_t run = part[tid] - s, trun = tpart[tid] - ts;
  for (uint32_t i = lo; i < hi; i++) {
    uint32_t v = bin_count[(uint64_t)i * BIN_PAD];
    bin_start[i] = run;
    tile_start[i] = trun;
    bin_cursor[(uint64_t)i * BIN_PAD] = 0;
    run += v;
    trun += (v + BIN_TILE - 1) / BIN_TILE;
  }
  if (tid == 1023) {
    bin_start[nbins] = part[1023];
    tile_start[nbins] = tpart[1023];
    *total_out = part[1023];  // number of sorted entries, for the host (pinned memory)
  }
}

// One workgroup takes PART_TILE slots of ONE window: its entries fall into that window's
// 2^(c-1-fine_bits) <= 1024 coarse bins only, so they can be grouped by bin in LDS and leave as
// contiguous runs (a tile of all windows at once would spread over every bin, a few entries each,
// and every 4-byte store would cost a 32-byte memory write).
constexpr uint32_t PART_TILE = 8192;
__global__ void __launch_bounds__(1024) k_part_scatter(MsmPlan plan, SortPlan sp, const uint32_t* __restrict__ dig,
                                                       const uint32_t* __restrict__ bin_start,
                                                       uint32_t* __restrict__ bin_cursor, uint32_t* __restrict__ ent_idx,
                                                       uint16_t* __restrict__ ent_fine) {
  __shared__ uint32_t cnt[1024];   // counts, then cursors into buf
  __shared__ uint32_t gdst[1024];  // (global position - LDS position) of each bin's run
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t buf[PART_TILE];
  __shared__ uint16_t bfine[PART_TILE];
  __shared__ uint16_t bbin[PART_TILE];
  const uint32_t tid = threadIdx.x;
  const uint32_t wj = blockIdx.y;  // m*W + j
  const int m = (int)(wj / (uint32_t)plan.W);
  const uint32_t wb = plan.table_stride ? (uint32_t)m : wj;                            // bucket set of this window
  const uint32_t tab_off = plan.table_stride * (wj - (uint32_t)m * (uint32_t)plan.W);  // this window's multiples
  const uint32_t nbw = 1u << (plan.c - 1 - sp.fine_bits);  // coarse bins per window
  const uint32_t fine_mask = (1u << sp.fine_bits) - 1u;
  const uint32_t t0 = blockIdx.x * PART_TILE;
  const uint32_t t1 = t0 + PART_TILE < plan.n ? t0 + PART_TILE : plan.n;
  const uint32_t* dw = dig + (uint64_t)wj * plan.n;
  cnt[tid] = 0;
  // the thread's PART_TILE / 1024 digit words stay in registers for both passes
  uint32_t dvr[PART_TILE / 1024];
#pragma unroll
  for (uint32_t k = 0; k < PART_TILE / 1024; k++) {
    const uint32_t t = t0 + tid + 1024u * k;
    dvr[k] = t < t1 ? dw[t] : 0u;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < PART_TILE / 1024; k++) {
    uint32_t d = dvr[k] & 0x7fffffffu;
    if (d) atomicAdd(&cnt[(d - 1) >> sp.fine_bits], 1u);
  }
  __syncthreads();
  const uint32_t v = cnt[tid];
  uint32_t total;
  const uint32_t loff = wg_excl_scan_1024(v, wsum, total);
  uint32_t g = 0;
  if (tid < nbw && v) {
    uint32_t b = wb * nbw + tid;
    g = bin_start[b] + atomicAdd(&bin_cursor[(uint64_t)b * BIN_PAD], v);
  }
  gdst[tid] = g - loff;
  cnt[tid] = loff;
  __syncthreads();
  // segment of the slot; slots advance through at most 3 segments
#pragma unroll
  for (uint32_t k = 0; k < PART_TILE / 1024; k++) {
    const uint32_t dv = dvr[k];
    const uint32_t d = dv & 0x7fffffffu;
    if (!d) continue;
    uint32_t r = t0 + tid + 1024u * k;
    int s = 0;
    while (s < 2 && r >= plan.seg[m][s].count) {
      r -= plan.seg[m][s].count;
      s++;
    }
    uint32_t lb = (d - 1) >> sp.fine_bits;
    uint32_t p = atomicAdd(&cnt[lb], 1u);
    buf[p] = (plan.seg[m][s].pt_off + r + tab_off) | (dv & 0x80000000u);
    bfine[p] = (uint16_t)((d - 1) & fine_mask);
    bbin[p] = (uint16_t)lb;
  }
  __syncthreads();
  for (uint32_t p = tid; p < total; p += 1024) {
    uint32_t pos = gdst[bbin[p]] + p;
    ent_idx[pos] = buf[p];
    ent_fine[pos] = bfine[p];
  }
}

// Second level: counting sort of every bin by fine bucket.  A bin is cut into tiles of up to
// BIN_TILE entries so that a bin holding a large share of all entries (the sparse top window, skewed
// scalars) is spread over many workgroups; counters live in LDS, one global atomic per
// (tile, non-empty fine bucket).
//   k_bin_count    per tile: LDS histogram -> bcount[bucket]
//   k_bin_offsets  per bin: scan -> offs[bucket]; bucket order by size; heavy-bucket lists
//   k_bin_place    per tile: reserve a slice of each bucket, place the entries -> sorted[]
__device__ __forceinline__ bool tile_lookup(const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ bin_start,
                                            uint32_t nbins, uint32_t t, uint32_t& bin, uint32_t& lo, uint32_t& hi) {
  if (t >= tile_start[nbins]) return false;
  uint32_t a = 0, b = nbins;  // largest a with tile_start[a] <= t
  while (b - a > 1) {
    uint32_t mid = (a + b) >> 1;
    if (tile_start[mid] <= t) a = mid; else b = mid;
  }
  bin = a;
  lo = bin_start[a] + (t - tile_start[a]) * BIN_TILE;
  uint32_t end = bin_start[a + 1];
  hi = lo + BIN_TILE < end ? lo + BIN_TILE : end;
  return true;
}

__global__ void __launch_bounds__(1024) k_bin_count(SortPlan sp, const uint32_t* __restrict__ bin_start,
                                                    const uint32_t* __restrict__ tile_start,
                                                    const uint16_t* __restrict__ ent_fine, uint32_t* __restrict__ bcount) {
  __shared__ uint32_t cnt[1024];
  const uint32_t tid = threadIdx.x, F = 1u << sp.fine_bits;
  uint32_t bin, lo, hi;
  if (!tile_lookup(tile_start, bin_start, sp.nbins, blockIdx.x, bin, lo, hi)) return;  // uniform per workgroup
  cnt[tid] = 0;
  __syncthreads();
  // the thread's BIN_TILE / 1024 loads go out together: one after the other (load, LDS atomic, load, ...) the loop
  // was a chain of memory latencies (335 -> 177 us at 2^24)
  uint32_t f[BIN_TILE / 1024];
#pragma unroll
  for (uint32_t k = 0; k < BIN_TILE / 1024; k++) {
    const uint32_t e = lo + tid + 1024u * k;
    f[k] = e < hi ? (uint32_t)ent_fine[e] : 0xffffffffu;
  }
#pragma unroll
  for (uint32_t k = 0; k < BIN_TILE / 1024; k++)
    if (f[k] != 0xffffffffu) atomicAdd(&cnt[f[k]], 1u);
  __syncthreads();
  uint32_t v = cnt[tid];
  if (tid < F && v) atomicAdd(&bcount[(uint64_t)bin * F + tid], v);
}

__global__ void __launch_bounds__(1024) k_bin_offsets(SortPlan sp, const uint32_t* __restrict__ bin_start,
                                                      uint32_t* __restrict__ bcount, uint32_t* __restrict__ offs,
                                                      uint32_t* __restrict__ order, HeavyLists hv) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t wsum[16];
  const uint32_t bin = blockIdx.x, tid = threadIdx.x;
  const uint32_t F = 1u << sp.fine_bits;
  const uint32_t bs = bin_start[bin], be = bin_start[bin + 1];
  uint32_t mine = tid < F ? bcount[(uint64_t)bin * F + tid] : 0u;
  if (tid < F) bcount[(uint64_t)bin * F + tid] = 0;  // becomes k_bin_place's cursor
  uint32_t bin_total;
  uint32_t start = bs + wg_excl_scan_1024(mine, wsum, bin_total);
  if (tid < F) offs[(uint64_t)bin * F + tid] = start;
  if (bin == gridDim.x - 1 && tid == 0) offs[(uint64_t)gridDim.x * F] = be;
  // a bucket with more than hv.threshold entries is cut into pieces of HEAVY_PIECE entries that whole
  // waves accumulate; single lanes skip it
  if (tid < F && mine > hv.threshold) {
    uint32_t np = (mine + hv.piece - 1) / hv.piece;
    uint32_t h = atomicAdd(&hv.counters[0], 1u);
    uint32_t p0 = atomicAdd(&hv.counters[1], np);
    hv.bucket[h] = bin * F + tid;
    hv.piece0[h] = p0;
    hv.npieces[h] = np;
    for (uint32_t i = 0; i < np; i++) {
      uint32_t lo = start + i * hv.piece;
      uint32_t hi = lo + hv.piece < start + mine ? lo + hv.piece : start + mine;
      hv.piece_lo[p0 + i] = lo;
      hv.piece_hi[p0 + i] = hi;
    }
  }
  __syncthreads();
  // order[]: this bin's buckets by decreasing size, so that the 64 lanes of a k_accum wave
  // walk lists of (nearly) equal length
  {
    const uint32_t key = 255u - (mine < 255u ? mine : 255u);
    part[tid] = 0;
    __syncthreads();
    if (tid < F) atomicAdd(&part[key], 1u);
    __syncthreads();
    uint32_t h = tid < 256 ? part[tid] : 0;
    __syncthreads();
    part[tid] = h;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
      uint32_t v = (tid >= d && tid < 256) ? part[tid - d] : 0;
      __syncthreads();
      if (tid < 256) part[tid] += v;
      __syncthreads();
    }
    uint32_t excl = tid < 256 ? part[tid] - h : 0;
    __syncthreads();
    if (tid < 256) part[tid] = excl;
    __syncthreads();
    if (tid < F) {
      uint32_t pos = atomicAdd(&part[key], 1u);
      order[(uint64_t)bin * F + pos] = bin * F + tid;
    }
  }
}

// The tile's entries are first grouped by bucket in LDS and then written out in index order, so
// that a wave's stores cover a few contiguous runs instead of 64 scattered words (scattered 4-byte
// stores cost a 32-byte memory write each: WRITE_SIZE was 6x the algorithmic bytes).
__global__ void __launch_bounds__(1024) k_bin_place(SortPlan sp, const uint32_t* __restrict__ bin_start,
                                                    const uint32_t* __restrict__ tile_start,
                                                    const uint32_t* __restrict__ ent_idx,
                                                    const uint16_t* __restrict__ ent_fine, const uint32_t* __restrict__ offs,
                                                    uint32_t* __restrict__ bcursor, uint32_t* __restrict__ sorted) {
  __shared__ uint32_t cnt[1024];    // counts, then cursors into buf
  __shared__ uint32_t gdst[1024];   // (global position - LDS position) of each bucket's run
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t buf[BIN_TILE];
  __shared__ uint16_t bfine[BIN_TILE];
  const uint32_t tid = threadIdx.x, F = 1u << sp.fine_bits;
  uint32_t bin, lo, hi;
  if (!tile_lookup(tile_start, bin_start, sp.nbins, blockIdx.x, bin, lo, hi)) return;
  cnt[tid] = 0;
  // the thread's BIN_TILE / 1024 fine-bucket numbers stay in registers for both passes
  uint32_t fr[BIN_TILE / 1024], ix[BIN_TILE / 1024];
#pragma unroll
  for (uint32_t k = 0; k < BIN_TILE / 1024; k++) {
    const uint32_t e = lo + tid + 1024u * k;
    fr[k] = e < hi ? (uint32_t)ent_fine[e] : 0xffffffffu;
    ix[k] = e < hi ? ent_idx[e] : 0u;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < BIN_TILE / 1024; k++)
    if (fr[k] != 0xffffffffu) atomicAdd(&cnt[fr[k]], 1u);
  __syncthreads();
  const uint32_t v = cnt[tid];
  uint32_t total;
  const uint32_t loff = wg_excl_scan_1024(v, wsum, total);  // first LDS slot of this bucket's run
  uint32_t g = 0;
  if (tid < F && v) {
    uint64_t bkt = (uint64_t)bin * F + tid;
    g = offs[bkt] + atomicAdd(&bcursor[bkt], v);  // this tile's slice of the bucket
  }
  gdst[tid] = g - loff;
  cnt[tid] = loff;
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < BIN_TILE / 1024; k++) {
    const uint32_t f = fr[k];
    if (f == 0xffffffffu) continue;
    uint32_t p = atomicAdd(&cnt[f], 1u);
    buf[p] = ix[k];
    bfine[p] = (uint16_t)f;
  }
  __syncthreads();
  const uint32_t cntn = hi - lo;
  for (uint32_t p = tid; p < cntn; p += 1024) sorted[gdst[bfine[p]] + p] = buf[p];
}

// sum over the first 2*FIRST lanes of a wave, result valid in lane 0
template <int BF, int FIRST = 32>
__device__ __forceinline__ Xyzz wave_sum_xyzz(Xyzz acc) {
  using E = Ec<BF>;
#pragma unroll 1
  for (int off = FIRST; off >= 1; off >>= 1) {
    Xyzz o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      o.x.l[i] = __shfl_down(acc.x.l[i], off, 64);
      o.y.l[i] = __shfl_down(acc.y.l[i], off, 64);
      o.zz.l[i] = __shfl_down(acc.zz.l[i], off, 64);
      o.zzz.l[i] = __shfl_down(acc.zzz.l[i], off, 64);
    }
    o.inf = __shfl_down((int)acc.inf, off, 64) != 0;
    acc = E::add(acc, o);
  }
  return acc;
}

// sums over aligned groups of 2*first lanes (first a power of two), result valid in each group's first lane
template <int BF>
__device__ __forceinline__ Xyzz group_sum_xyzz(Xyzz acc, int first) {
  using E = Ec<BF>;
#pragma unroll 1
  for (int off = first; off >= 1; off >>= 1) {
    Xyzz o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      o.x.l[i] = __shfl_down(acc.x.l[i], off, 64);
      o.y.l[i] = __shfl_down(acc.y.l[i], off, 64);
      o.zz.l[i] = __shfl_down(acc.zz.l[i], off, 64);
      o.zzz.l[i] = __shfl_down(acc.zzz.l[i], off, 64);
    }
    o.inf = __shfl_down((int)acc.inf, off, 64) != 0;
    acc = E::add(acc, o);
  }
  return acc;
}

// ---- quad-cooperative addition -----------------------------------------------------------------------------------
// The bucket reduction of a small MSM (the prover's short rounds, 2^16 points and below) is a CHAIN of some twenty
// dependent XYZZ additions on a handful of lanes: a lone wave issues an instruction every five clocks, so one addition
// (fourteen multiplications one after the other) is eight microseconds however idle the chip is.  Here the four lanes of
// a quad hold the SAME two operands and share the multiplications of ONE addition: four rounds of one multiplication
// per lane (add-2008-s has depth four), results passed round the quad by DPP moves (one VALU slot per limb, no LDS).
//   round 1:  U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1          P = U2 - U1, R = S2 - S1
//   round 2:  PP = P P    | RR = R R    | ZZ1 ZZ2      | ZZZ1 ZZZ2
//   round 3:  PPP = P PP  | Q = U1 PP   | ZZ3 = (ZZ1 ZZ2) PP | (idle)           X3 = RR - PPP - 2 Q
//   round 4:  R (Q - X3)  | S1 PPP      | (idle)       | ZZZ3 = (ZZZ1 ZZZ2) PPP  Y3 = difference of the first two
// About a third of the instructions of Ec::add on the critical path.  Every lane of the quad returns the same sum.
// All four lanes of a quad must call together with identical operands (quads are aligned groups of four lanes).
template <int J>
__device__ __forceinline__ Fe fe_quad_bcast(const Fe& a) {
  Fe r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.l[k] = __builtin_amdgcn_update_dpp(a.l[k], a.l[k], J * 0x55, 0xf, 0xf, false);  // quad_perm:[J,J,J,J]
  return r;
}
__device__ __forceinline__ Fe fe_sel4(uint32_t q, const Fe& a0, const Fe& a1, const Fe& a2, const Fe& a3) {
  Fe r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int32_t lo = (q & 1u) ? a1.l[k] : a0.l[k], hi = (q & 1u) ? a3.l[k] : a2.l[k];
    r.l[k] = (q & 2u) ? hi : lo;
  }
  return r;
}
template <int BF>
__device__ __forceinline__ Xyzz quad_add(const Xyzz& a, const Xyzz& b) {
  using F = Fp<BF>;
  using E = Ec<BF>;
  if (a.inf) return b;  // the operands are identical across the quad, so whole quads take these branches together
  if (b.inf) return a;
  const uint32_t q = threadIdx.x & 3u;
  const Fe r1 = F::mul(fe_sel4(q, a.x, b.x, a.y, b.y), fe_sel4(q, b.zz, a.zz, b.zzz, a.zzz));
  const Fe U1 = fe_quad_bcast<0>(r1), U2 = fe_quad_bcast<1>(r1), S1 = fe_quad_bcast<2>(r1), S2 = fe_quad_bcast<3>(r1);
  const Fe P = F::sub(U2, U1), R = F::sub(S2, S1);
  if (F::is_zero(P)) {
    if (F::is_zero(R)) return E::dbl(a);
    return E::identity();
  }
  const Fe r2 = F::mul(fe_sel4(q, P, R, a.zz, a.zzz), fe_sel4(q, P, R, b.zz, b.zzz));
  const Fe PP = fe_quad_bcast<0>(r2), RR = fe_quad_bcast<1>(r2);
  const Fe r3 = F::mul(fe_sel4(q, P, U1, r2, P), PP);
  const Fe PPP = fe_quad_bcast<0>(r3), Q = fe_quad_bcast<1>(r3);
  Xyzz c;
  c.zz = fe_quad_bcast<2>(r3);
  c.x = F::carry(F::sub(F::sub(F::sub(RR, PPP), Q), Q));
  const Fe QX = F::sub(Q, c.x);
  const Fe r4 = F::mul(fe_sel4(q, R, S1, R, r2), fe_sel4(q, QX, PPP, QX, PPP));
  c.y = F::carry(F::sub(fe_quad_bcast<0>(r4), fe_quad_bcast<1>(r4)));
  c.zzz = fe_quad_bcast<3>(r4);
  c.inf = false;
  return c;
}

// lane l takes the value of lane src(l) (any lane map; LDS crossbar)
template <int BF>
__device__ __forceinline__ Xyzz xyzz_from_lane(const Xyzz& a, int src) {
  Xyzz o;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    o.x.l[i] = __shfl(a.x.l[i], src, 64);
    o.y.l[i] = __shfl(a.y.l[i], src, 64);
    o.zz.l[i] = __shfl(a.zz.l[i], src, 64);
    o.zzz.l[i] = __shfl(a.zzz.l[i], src, 64);
  }
  o.inf = __shfl((int)a.inf, src, 64) != 0;
  return o;
}
// Sum of the values held by lanes [0, cnt) of a wave (cnt a power of two, wave-uniform); the result is valid in lanes 0..3.
// Levels with more than sixteen additions run one addition per lane (Ec::add); then value j moves to quad j and the
// remaining levels -- at most sixteen additions, four lanes each -- run quad-cooperatively.  Every lane must call.
template <int BF>
__device__ __forceinline__ Xyzz tree_sum_quad(Xyzz v, int cnt) {
  using E = Ec<BF>;
  const int lane = (int)(threadIdx.x & 63u);
#pragma unroll 1
  while (cnt > 16) {
    cnt >>= 1;
    v = E::add(v, xyzz_from_lane<BF>(v, lane + cnt < 64 ? lane + cnt : lane));
  }
  v = xyzz_from_lane<BF>(v, lane >> 2);
#pragma unroll 1
  for (int off = 2 * cnt; off >= 4; off >>= 1) v = quad_add<BF>(v, xyzz_from_lane<BF>(v, lane + off < 64 ? lane + off : lane));
  return v;
}

// Bucket accumulation.  BF = base field id.  Persistent waves: the grid is sized to the chip
// (2 waves per SIMD) and every wave pulls work items from a shared counter until none is left
// (bounded loop; every wave reaches item >= total and leaves).  Items, longest first:
//   * the pieces of heavy buckets (skewed scalars, or the top window whose few significant bits
//     leave it a handful of buckets): the 64 lanes stride through <= HEAVY_PIECE entries and
//     their partial sums are combined across the wave (tree_sum_quad) -> psum[piece];
//   * chunks of 64 ordinary buckets, one per lane: chunk k is the (k / nbins)-th group of 64
//     buckets, in decreasing size, of bin k % nbins (k_bin_offsets' order[]), so lanes of a wave
//     walk lists of near-equal length and the kernel's tail is made of the shortest lists.
//     When there are far fewer buckets than lanes (fixed-base table mode: one bucket set for all windows,
//     lists of a hundred entries and more) G = 2^glog lanes share a bucket: a chunk is 64 / G buckets, a
//     lane strides its bucket's list by G and the G partial sums are added up by lane shuffles.
#ifndef BP_ACCUM_WAVES
#define BP_ACCUM_WAVES 2  // persistent waves per SIMD (tuning knob; also the launch bound)
#endif
template <int BF>
__global__ void __launch_bounds__(256, BP_ACCUM_WAVES) k_accum(const uint32_t* __restrict__ points, const uint32_t* __restrict__ offs,
                                               const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ order,
                                               uint32_t* __restrict__ bsum, uint32_t nbuckets, uint32_t nbins,
                                               int fine_bits, HeavyLists hv, uint32_t* __restrict__ work_counter, int glog) {
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t per = 64u >> glog;  // buckets per chunk
  const uint32_t sub = lane & ((1u << glog) - 1u), slot = lane >> glog;
  const uint32_t nchunks = (nbuckets + per - 1u) / per;
  const uint32_t npieces = hv.counters[1];
  const uint32_t total = npieces + nchunks;
  const uint32_t F = 1u << fine_bits;
  for (uint32_t it = 0; it <= total; it++) {  // bounded: a wave can never pull more than every item
    uint32_t k = 0;
    if (lane == 0) k = atomicAdd(work_counter, 1u);
    k = __builtin_amdgcn_readfirstlane(k);
    if (k >= total) break;
    // no lane may leave the loop body early: the counter is read by lane 0 for the whole wave
    const bool piece = k < npieces;
    bool act = true;
    uint32_t b = 0, lo = 0, hi = 0, step = 1;
    if (piece) {
      lo = hv.piece_lo[k] + lane;
      hi = hv.piece_hi[k];
      step = 64;
    } else {
      uint32_t kc = k - npieces, gid;
      if (F >= per) {
        uint32_t pos = kc / nbins, bin = kc - pos * nbins;
        gid = bin * F + pos * per + slot;
      } else {
        gid = kc * per + slot;
      }
      act = gid < nbuckets && (F < per || (kc / nbins) * per + slot < F);
      if (act) {
        b = order[gid];
        lo = offs[b];
        hi = offs[b + 1];
        act = hi - lo <= hv.threshold;  // longer lists were cut into pieces
        lo += sub;
        step = 1u << glog;
      }
      if (!act) hi = lo;
    }
    Xyzz acc = E::identity();
    // software pipeline: the index two entries ahead and the point one entry ahead are in flight
    // while the current mixed add runs.  The point is unpacked at the BOTTOM of the iteration that
    // loaded it: nothing at the top of the loop body depends on a pending load, so the only wait
    // sits behind the mixed add (with the raw words carried across the back edge the compiler put
    // a wait for the just-issued index load at the top: a full memory latency per entry).
    uint32_t ent = lo < hi ? sorted[lo] : 0u;
    uint32_t ent_n = lo + step < hi ? sorted[lo + step] : 0u;
    Aff q;
    {
      const uint4* pp = reinterpret_cast<const uint4*>(points + 16ull * (ent & 0x7fffffffu));
      const uint4 r0 = pp[0], r1 = pp[1], r2 = pp[2], r3 = pp[3];
      uint32_t w[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
      q = IO::load_dev_affine(w);
      if (ent >> 31) q = E::neg(q);
    }
    for (uint32_t e = lo; e < hi; e += step) {
      const uint32_t ent_nn = e + 2 * step < hi ? sorted[e + 2 * step] : 0u;
      const uint4* pp = reinterpret_cast<const uint4*>(points + 16ull * (ent_n & 0x7fffffffu));
      const uint4 n0 = pp[0], n1 = pp[1], n2 = pp[2], n3 = pp[3];
      E::madd(acc, q);
      uint32_t w[16] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
      q = IO::load_dev_affine(w);
      if (ent_n >> 31) q = E::neg(q);
      ent_n = ent_nn;
    }
    if (piece) {  // wave-uniform
      acc = tree_sum_quad<BF>(acc, 64);
      if (lane == 0) IO::store_dev_xyzz(hv.psum + 32ull * k, acc);
    } else {
      if (glog) acc = group_sum_xyzz<BF>(acc, 1 << (glog - 1));  // wave-uniform; every lane takes part
      if (act && sub == 0) IO::store_dev_xyzz(bsum + 32ull * b, acc);
    }
  }
}

// bsum[heavy bucket] = sum of its pieces; one wave per heavy bucket (grid-stride)
template <int BF>
__global__ void __launch_bounds__(256, 2) k_heavy_merge(HeavyLists hv, uint32_t* __restrict__ bsum) {
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t nheavy = hv.counters[0];
  for (uint32_t h = wave; h < nheavy; h += nwaves) {
    const uint32_t p0 = hv.piece0[h], np = hv.npieces[h];
    Xyzz acc = E::identity();
    for (uint32_t i = lane; i < np; i += 64) acc = E::add(acc, IO::load_dev_xyzz(hv.psum + 32ull * (p0 + i)));
    acc = tree_sum_quad<BF>(acc, 64);
    if (lane == 0) IO::store_dev_xyzz(bsum + 32ull * hv.bucket[h], acc);
  }
}

// Bucket reduction (running-sum trick of msm.rs:167-171, restructured for depth): the window sum
// sum_k (k+1) B_k is split by runs of S = 2^s consecutive buckets.  For run r (window-local) with
// first bucket k0 = r S:
//     sum_{k in run} (k+1) B_k = T_r + r S * S_r,   T_r = sum (k-k0+1) B_k,   S_r = sum B_k
// k_reduce1: one lane per run, 2S additions, writes (T_r, S_r).
// k_reduce2: sum_r r S_r = sum_i 2^i U_i with U_i = sum of S_r over runs whose index has bit i set;
//            one workgroup per (window, component) tree-sums P = sum_r T_r or one U_i.
// The host folds the components into its Horner pass at bit positions c w (P, in two halves) and
// c w + s + i (U_i).
template <int BF>
__global__ void __launch_bounds__(256) k_reduce1(const uint32_t* __restrict__ bsum, uint32_t* __restrict__ partial,
                                                 uint32_t nruns, int s) {
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nruns) return;
  uint32_t S = 1u << s;
  uint32_t first = r << s;
  Xyzz running = E::identity(), total = E::identity();
  for (int k = (int)S - 1; k >= 0; k--) {
    Xyzz bk = IO::load_dev_xyzz(bsum + 32ull * (first + k));
    running = E::add(running, bk);
    total = E::add(total, running);
  }
  IO::store_dev_xyzz(partial + 64ull * r, total);
  IO::store_dev_xyzz(partial + 64ull * r + 32, running);
}

// k_reduce1 for few runs (a small MSM): one QUAD per run, the chain of 2 S additions runs at the quad-cooperative latency.
template <int BF>
__global__ void __launch_bounds__(256) k_reduce1_quad(const uint32_t* __restrict__ bsum, uint32_t* __restrict__ partial,
                                                      uint32_t nruns, int s) {
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  const uint32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  if (r >= nruns) return;  // whole quads leave together
  const uint32_t S = 1u << s, first = r << s;
  Xyzz running = E::identity(), total = E::identity();
  for (int k = (int)S - 1; k >= 0; k--) {
    const Xyzz bk = IO::load_dev_xyzz(bsum + 32ull * (first + k));  // the four lanes read the same words
    running = quad_add<BF>(running, bk);
    total = quad_add<BF>(total, running);
  }
  if ((threadIdx.x & 3u) == 0) {
    IO::store_dev_xyzz(partial + 64ull * r, total);
    IO::store_dev_xyzz(partial + 64ull * r + 32, running);
  }
}

// grid = (2 + nbits, windows, ceil(slices / 4)); components 0 and 1 = the two halves of P, component
// 2+i = U_i.  A component sums per/2 values; it is cut into `slices` slices, one wave each: lanes
// first (every lane busy), then the wave, then the (up to) four waves of the workgroup through LDS --
// the addition code is already in the instruction cache here, whereas a separate small kernel for that
// last step pays more in cold instruction fetches than in arithmetic.  With slices <= 4 the result
// (ark-form XYZZ) goes to out; otherwise one sum per workgroup goes to tmp and the output's last workgroup adds them
// up (tickets).
template <int BF>
__global__ void __launch_bounds__(256) k_reduce2(const uint32_t* __restrict__ partial, uint32_t* tmp,
                                                 uint32_t* __restrict__ out, uint32_t per, uint32_t slices,
                                                 uint32_t* __restrict__ tickets) {
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  __shared__ int32_t xs[4][37];
  const uint32_t comp = blockIdx.x, w = blockIdx.y, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t slice = blockIdx.z * 4 + wave;
  const uint32_t half = per >> 1;  // per is a power of two; per = 1 leaves component 1.. empty
  Xyzz acc = E::identity();
  if (slice < slices) {
    if (per == 1) {
      if (comp == 0 && lane == 0) acc = IO::load_dev_xyzz(partial + 64ull * w);
    } else {
      const uint32_t len = half / slices, j0 = slice * len;  // slices <= half, both powers of two
      for (uint32_t j = j0 + lane; j < j0 + len; j += 64) {
        uint32_t r, which;
        if (comp < 2) {
          r = comp * half + j;
          which = 0;  // T_r
        } else {
          uint32_t i = comp - 2;  // runs whose index has bit i set
          r = ((j >> i) << (i + 1)) | (1u << i) | (j & ((1u << i) - 1u));
          which = 32;  // S_r
        }
        acc = E::add(acc, IO::load_dev_xyzz(partial + 64ull * ((uint64_t)w * per + r) + which));
      }
    }
  }
  acc = tree_sum_quad<BF>(acc, 64);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      xs[wave][i] = acc.x.l[i];
      xs[wave][9 + i] = acc.y.l[i];
      xs[wave][18 + i] = acc.zz.l[i];
      xs[wave][27 + i] = acc.zzz.l[i];
    }
    xs[wave][36] = acc.inf ? 1 : 0;
  }
  __syncthreads();
  if (wave != 0) return;
  Xyzz v = E::identity();
  if (lane < 4) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      v.x.l[i] = xs[lane][i];
      v.y.l[i] = xs[lane][9 + i];
      v.zz.l[i] = xs[lane][18 + i];
      v.zzz.l[i] = xs[lane][27 + i];
    }
    v.inf = xs[lane][36] != 0;
  }
  v = tree_sum_quad<BF>(v, 4);
  const uint64_t o = (uint64_t)w * gridDim.x + comp;
  if (lane == 0) {
    if (gridDim.z == 1)
      IO::store_ark_xyzz(out + 32ull * o, v);
    else
      IO::store_dev_xyzz(tmp + 32ull * (o * gridDim.z + blockIdx.z), v);
  }
  if (gridDim.z == 1 || tickets == nullptr) return;
  // The last workgroup of this output (a ticket per output, zeroed by the host with the sort's counters) adds up the
  // sums of all gridDim.z (a power of two, at most 64) of them.  Release / acquire by device-scope fences around the
  // ticket: the other workgroups' stores may sit in another XCD's L2.
  uint32_t t = 0;
  if (lane == 0) {
    __threadfence();
    t = atomicAdd(&tickets[o], 1u);
  }
  t = __builtin_amdgcn_readfirstlane(t);
  if (t != gridDim.z - 1) return;  // wave-uniform
  __threadfence();
  Xyzz a = E::identity();
  if (lane < gridDim.z) a = IO::load_dev_xyzz(tmp + 32ull * (o * gridDim.z + lane));
  a = tree_sum_quad<BF>(a, (int)gridDim.z);
  if (lane == 0) IO::store_ark_xyzz(out + 32ull * o, a);
}

// ---------------------------------------------------------------------------
// Format conversion kernels
// ---------------------------------------------------------------------------
// ark jacobian (96 B) or ark affine (64 B) -> dev affine (64 B)
// one staged point (fmt 0: ark jacobian, 96 B; fmt 1: ark affine, 64 B) -> device affine
template <int BF>
__device__ __forceinline__ Aff point_in(const uint32_t* __restrict__ p, int fmt) {
  using F = Fp<BF>;
  using IO = PointIo<BF>;
  uint32_t wx[8], wy[8], wz[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    wx[k] = p[k];
    wy[k] = p[8 + k];
  }
  Aff a;
  if (fmt == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) wz[k] = p[16 + k];
    bool z_one = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      z_one = z_one && wz[2 * k] == (uint32_t)F::C::H_ONE[k] && wz[2 * k + 1] == (uint32_t)(F::C::H_ONE[k] >> 32);
    }
    a.inf = IO::all_zero(wz);
    a.x = F::load_ark(wx);
    a.y = F::load_ark(wy);
    if (!z_one && !a.inf) {
      Fe z = F::load_ark(wz);
      Fe zi = F::inv(z);
      Fe zi2 = F::sqr(zi);
      a.x = F::mul(a.x, zi2);
      a.y = F::mul(a.y, F::mul(zi2, zi));
    }
  } else {
    a.inf = IO::all_zero(wx) && IO::all_zero(wy);
    a.x = F::load_ark(wx);
    a.y = F::load_ark(wy);
  }
  return a;
}
template <int BF>
__global__ void __launch_bounds__(256) k_points_in(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                   uint32_t n, int fmt) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PointIo<BF>::store_dev_affine(out + 16ull * i, point_in<BF>(in + (uint64_t)i * (fmt == 0 ? 24 : 16), fmt));
}

// A whole segment of a chunked upload at once (bpgpu.hip, pt_upload_begin with a deferred phase): the staging area
// holds the segment's chunks `chunk` points apart at 96 B a point, each chunk in the format its staging thread found
// (fmt[c], written to pinned host memory before the chunk's DMA was enqueued): rows of 64 B or of 96 B from the
// chunk's start.
template <int BF>
__global__ void __launch_bounds__(256) k_points_in_chunked(const uint32_t* __restrict__ stage, uint32_t* __restrict__ out,
                                                           uint32_t n, uint32_t chunk, const uint8_t* __restrict__ fmt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t c = i / chunk, r = i - c * chunk;
  const int f = fmt[c];  // uniform over a workgroup when 256 divides `chunk`
  PointIo<BF>::store_dev_affine(out + 16ull * i, point_in<BF>(stage + 24ull * c * chunk + (uint64_t)r * (f == 0 ? 24 : 16), f));
}

// The same conversion for rows the host has already found affine (x, y in ark form, 64 B): no inversion path, 32
// registers, whole 256-thread workgroups -- what the pipelined uploads launch per chunk.  (Shaped to fit beside a bucket
// accumulation -- few registers, no LDS, raised wave priority -- it still does not finish before that kernel ends:
// kernel trace of a 2^20-element proof; the uploader therefore keeps conversions off the DMA streams, bpgpu.hip.)
template <int BF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_points_in_affine(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n) {
  using F = Fp<BF>;
  using IO = PointIo<BF>;
  __builtin_amdgcn_s_setprio(3);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* p = reinterpret_cast<const uint4*>(in + 16ull * i);
  const uint4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
  const uint32_t wx[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  const uint32_t wy[8] = {r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
  Aff a;
  a.inf = IO::all_zero(wx) && IO::all_zero(wy);
  a.x = F::load_ark(wx);
  a.y = F::load_ark(wy);
  IO::store_dev_affine(out + 16ull * i, a);
}

// dev affine -> ark jacobian with Z = 1, identity = (0, 1, 0)  (ark-ec's zero())
template <int BF>
__global__ void __launch_bounds__(256) k_points_out(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                    uint32_t n) {
  using F = Fp<BF>;
  using IO = PointIo<BF>;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Aff a = IO::load_dev_affine(in + 16ull * i);
  uint32_t wx[8], wy[8];
  F::store_ark(a.x, wx);
  F::store_ark(a.y, wy);
  uint32_t* o = out + 24ull * i;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t one_lo = (uint32_t)F::C::H_ONE[k], one_hi = (uint32_t)(F::C::H_ONE[k] >> 32);
    o[2 * k] = a.inf ? 0u : wx[2 * k];
    o[2 * k + 1] = a.inf ? 0u : wx[2 * k + 1];
    o[8 + 2 * k] = a.inf ? one_lo : wy[2 * k];
    o[8 + 2 * k + 1] = a.inf ? one_hi : wy[2 * k + 1];
    o[16 + 2 * k] = a.inf ? 0u : one_lo;
    o[16 + 2 * k + 1] = a.inf ? 0u : one_hi;
  }
}

// ---------------------------------------------------------------------------
// Scalar-vector kernels (scalar field SF); vectors are in ark form (value*2^256)
// ---------------------------------------------------------------------------
// a (|value| < 100p, any lazy limbs)  ->  carried limbs, value in (-2^130, 2^254 + 2^130):
// subtracts floor(value / 2^254) * p so that long sums can go on without growing
template <int SF>
__device__ __forceinline__ Fe fe_partial_reduce(const Fe& a) {
  using F = Fp<SF>;
  Fe r = F::carry(a);
  int32_t k = r.l[8] >> 22;  // floor(value / 2^254), |k| <= 128
  int64_t cy = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    int64_t s = (int64_t)r.l[i] - (int64_t)k * (int64_t)F::C::P29[i] + cy;
    r.l[i] = (int32_t)((uint32_t)s & MASK29);
    cy = s >> 29;
  }
#pragma unroll
  for (int i = 5; i < 8; i++) {
    int32_t s = r.l[i] + (int32_t)cy;
    r.l[i] = s & (int32_t)MASK29;
    cy = s >> 29;
  }
  r.l[8] = r.l[8] - k * (int32_t)F::C::P29[8] + (int32_t)cy;
  return r;
}

struct IpPair {
  uint32_t a_off, b_off;  // element offsets of a into sa and of b into sb
};

// partial[blockIdx.y][blockIdx.x] = sum over this block's slice of a[i]*b[i] (device form of the
// *ark-form product*, see k_ip_final).  Up to two inner products per launch (the round's
// <a_lo,b_hi> and <a_hi,b_lo>, bp_iter.rs:73,80).
template <int SF>
__global__ void __launch_bounds__(256) k_ip_partial(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ sb, IpPair p0,
                                                    IpPair p1, uint32_t n, uint32_t* __restrict__ partial) {
  using F = Fp<SF>;
  __shared__ int32_t sh[256 * 9];
  IpPair pr = blockIdx.y == 0 ? p0 : p1;
  const uint32_t* a = sa + 8ull * pr.a_off;
  const uint32_t* b = sb + 8ull * pr.b_off;
  uint32_t tid = threadIdx.x;
  Fe acc = F::zero();
  // 64-bit index: n may be anything up to 2^32 - 1 and i + gridDim.x * 256 must not wrap
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + tid; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint4* pa = reinterpret_cast<const uint4*>(a + 8ull * i);
    const uint4* pb = reinterpret_cast<const uint4*>(b + 8ull * i);
    uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    uint32_t wa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    uint32_t wb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    // (a 2^256)(b 2^256)/2^261 = ab 2^251
    Fe pdt = F::mul(F::unpack(wa), F::unpack(wb));
    acc = fe_partial_reduce<SF>(F::add(acc, pdt));
  }
#pragma unroll
  for (int k = 0; k < 9; k++) sh[k * 256 + tid] = acc.l[k];
  for (uint32_t d = 128; d >= 1; d >>= 1) {
    __syncthreads();
    if (tid < d) {
      Fe o;
#pragma unroll
      for (int k = 0; k < 9; k++) o.l[k] = sh[k * 256 + tid + d];
      acc = fe_partial_reduce<SF>(F::add(acc, o));
#pragma unroll
      for (int k = 0; k < 9; k++) sh[k * 256 + tid] = acc.l[k];
    }
  }
  if (tid == 0) {
    uint32_t w[8];
    F::store_packed(acc, w);
    uint32_t* o = partial + 8ull * (blockIdx.y * gridDim.x + blockIdx.x);
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = w[k];
  }
}

// out[y] = (sum_x partial[y][x]) * 2^5   i.e. ab 2^251 -> ab 2^256 (ark form); one workgroup per pair
template <int SF>
__global__ void __launch_bounds__(256) k_ip_final(const uint32_t* __restrict__ partial, uint32_t nparts,
                                                  uint32_t* __restrict__ out0, uint32_t* __restrict__ out1) {
  using F = Fp<SF>;
  __shared__ int32_t sh[256 * 9];
  const uint32_t tid = threadIdx.x;
  Fe acc = F::zero();
  for (uint32_t i = tid; i < nparts; i += 256) {
    const uint4* pp = reinterpret_cast<const uint4*>(partial + 8ull * (blockIdx.x * nparts + i));
    uint4 p0 = pp[0], p1 = pp[1];
    uint32_t w[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    acc = fe_partial_reduce<SF>(F::add(acc, F::unpack(w)));
  }
#pragma unroll
  for (int k = 0; k < 9; k++) sh[k * 256 + tid] = acc.l[k];
  for (uint32_t d = 128; d >= 1; d >>= 1) {
    __syncthreads();
    if (tid < d) {
      Fe o;
#pragma unroll
      for (int k = 0; k < 9; k++) o.l[k] = sh[k * 256 + tid + d];
      acc = fe_partial_reduce<SF>(F::add(acc, o));
#pragma unroll
      for (int k = 0; k < 9; k++) sh[k * 256 + tid] = acc.l[k];
    }
  }
  if (tid == 0) {
    // montmul(x 2^251, 2^266) = x 2^256
    Fe r = F::mul(acc, F::from_const(F::C::C256TO261));
    uint32_t w[8];
    F::store_packed(r, w);
    uint32_t* o = blockIdx.x == 0 ? out0 : out1;
#pragma unroll
    for (int k = 0; k < 8; k++) o[k] = w[k];
  }
}

struct FoldScalars {
  uint32_t lo[8], hi[8];  // ark form
};

// out[i] = s_lo v[i] + s_hi v[i+n]  (bp_iter.rs:88-95).  blockIdx.y selects one of two
// vectors (a with (x, x^-1) and b with (x^-1, x) in the same launch).
template <int SF>
__global__ void __launch_bounds__(256) k_fold_scalars(uint32_t* __restrict__ v0, uint32_t* __restrict__ o0,
                                                      FoldScalars s0, uint32_t* __restrict__ v1,
                                                      uint32_t* __restrict__ o1, FoldScalars s1, uint32_t n) {
  using F = Fp<SF>;
  const uint32_t* v = blockIdx.y == 0 ? v0 : v1;
  uint32_t* o = blockIdx.y == 0 ? o0 : o1;
  // device form of the two multipliers, once per lane: montmul(v 2^256, s 2^261) = vs 2^256, ark form again
  // (canonical: the products' sum then lies in (-p, p/16] and one conditional addition canonicalises it)
  Fe slo, shi;
  if (blockIdx.y == 0) {
    slo = F::canon(F::load_ark(s0.lo));
    shi = F::canon(F::load_ark(s0.hi));
  } else {
    slo = F::canon(F::load_ark(s1.lo));
    shi = F::canon(F::load_ark(s1.hi));
  }
  // `o` may alias `v`: every output reads only its own two inputs before it is written
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint4* pl = reinterpret_cast<const uint4*>(v + 8ull * i);
    const uint4* ph = reinterpret_cast<const uint4*>(v + 8ull * (i + n));
    uint4 l0 = pl[0], l1 = pl[1], h0 = ph[0], h1 = ph[1];
    uint32_t wl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    uint32_t wh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    Fe r = F::mul_add2(F::unpack(wl), slo, F::unpack(wh), shi);  // one reduction for the two products
    uint32_t w[8];
    F::pack(F::canon_small(r), w);
    uint4* po = reinterpret_cast<uint4*>(o + 8ull * i);
    po[0] = make_uint4(w[0], w[1], w[2], w[3]);
    po[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// ---------------------------------------------------------------------------
// Deferred generator folding for the short rounds of the prover.  Once the vectors are
// short, one lane's 128-doubling ladder (a millisecond) costs more than an MSM over the generators
// as they were when folding stopped ("frozen", n0 of each of A and B).  The current generator i
// (vector length 2n) is then  A_i = sum_{j = i mod 2n} wA_j A0_j  with wA_j the product of the
// challenges applied so far, so the round's L and R (bp_iter.rs:68-81) are MSMs over A0, B0 with
// scalars a_{.} wA_j / b_{.} wB_j, half of them zero.  Only L, R are proof outputs, so the folded
// generators are never needed (bpgpu_ipa_export replays the point folds if it is called).
// Weights are kept in device form (value * 2^261), so montmul(ark scalar, weight) is an ark scalar.
// ---------------------------------------------------------------------------
// weights start at the common factors carried by the session: w[0, half) = m.lo, w[half, 2 half) = m.hi
template <int SF>
__global__ void __launch_bounds__(256) k_frozen_init(uint32_t* __restrict__ w, uint32_t half, FoldScalars m) {
  using F = Fp<SF>;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * half) return;
  uint32_t o[8];
  F::store_packed(j < half ? F::load_ark(m.lo) : F::load_ark(m.hi), o);
#pragma unroll
  for (int k = 0; k < 8; k++) w[8ull * j + k] = o[k];
}

// after a fold with challenge x of vectors of length 2n:  A'_i = x^-1 A_i + x A_{i+n},  B'_i = x B_i + x^-1 B_{i+n}
template <int SF>
__global__ void __launch_bounds__(256) k_frozen_update(uint32_t* __restrict__ wA, uint32_t* __restrict__ wB, uint32_t n0,
                                                       uint32_t n, FoldScalars xs /* lo = x, hi = x^-1 */) {
  using F = Fp<SF>;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n0) return;
  const bool low = (j & (2 * n - 1)) < n;
  Fe x = F::load_ark(xs.lo), xi = F::load_ark(xs.hi);
  uint32_t a[8], b[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    a[k] = wA[8ull * j + k];
    b[k] = wB[8ull * j + k];
  }
  Fe na = F::mul(F::unpack(a), low ? xi : x), nb = F::mul(F::unpack(b), low ? x : xi);
  F::store_packed(na, a);
  F::store_packed(nb, b);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    wA[8ull * j + k] = a[k];
    wB[8ull * j + k] = b[k];
  }
}

// scalars of the round's two MSMs over the frozen generators; vectors a, b have length 2n.
// out layout (ark form): [L over A0 | L over B0 | R over A0 | R over B0], n0 each.
template <int SF>
__global__ void __launch_bounds__(256) k_frozen_scalars(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                        const uint32_t* __restrict__ wA, const uint32_t* __restrict__ wB,
                                                        uint32_t* __restrict__ out, uint32_t n0, uint32_t n) {
  using F = Fp<SF>;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n0) return;
  const uint32_t i = j & (2 * n - 1);
  const bool low = i < n;
  // L = msm(A[n..], a[..n]) + msm(B[..n], b[n..]);  R = msm(A[..n], a[n..]) + msm(B[n..], b[..n])
  const uint32_t ia = low ? i + n : i - n, ib = ia;
  uint32_t sa[8], sb[8], ua[8], ub[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    sa[k] = a[8ull * ia + k];
    sb[k] = b[8ull * ib + k];
    ua[k] = wA[8ull * j + k];
    ub[k] = wB[8ull * j + k];
  }
  uint32_t pa[8], pb[8];
  F::store_packed(F::mul(F::unpack(sa), F::unpack(ua)), pa);
  F::store_packed(F::mul(F::unpack(sb), F::unpack(ub)), pb);
  uint32_t* la = out + 8ull * j;
  uint32_t* lb = out + 8ull * (n0 + j);
  uint32_t* ra = out + 8ull * (2ull * n0 + j);
  uint32_t* rb = out + 8ull * (3ull * n0 + j);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    la[k] = low ? 0u : pa[k];  // A_j with j mod 2n >= n pairs with a[i - n] in L
    ra[k] = low ? pa[k] : 0u;  // A_j with j mod 2n <  n pairs with a[i + n] in R
    lb[k] = low ? pb[k] : 0u;  // B_j with j mod 2n <  n pairs with b[i + n] in L
    rb[k] = low ? 0u : pb[k];  // B_j with j mod 2n >= n pairs with b[i - n] in R
  }
}

// ---------------------------------------------------------------------------
// Generator fold (base field BF = curve id):  out[i] = s_lo * g[i] + s_hi * g[i+n]
// (bp_iter.rs:97-104).  Every lane uses the same two scalars, so they are recoded once on the
// host and the whole grid follows one branch-uniform schedule:
//   * GLV: the curve has the endomorphism phi(x, y) = (zeta x, y) = [lambda](x, y), so
//     s = k1 + k2 lambda with |k1|, |k2| < 2^129 halves the doublings:
//         s_lo P + s_hi Q = k1 P + k2 phi(P) + l1 Q + l2 phi(Q)
//   * each pair (|k1|, |k2|), (|l1|, |l2|) is written in joint sparse form (half the positions
//     are zero); signs of the halves are applied to the base points.
// 4-bit codes, position 0 = least significant:  code & 7: 0 none, 1 first base, 2 second base,
// 3 first + second, 4 first - second;  code & 8: negate.
// ---------------------------------------------------------------------------
// The recoded multipliers of a fold launch travel as kernel arguments (two plans, one per generator
// vector: at most 440 bytes), so a launch needs neither a device buffer nor a synchronisation.
template <class T>
struct Plans2 {
  T p[2];
};

struct FoldPlan {
  uint32_t codes_p[17];  // pair (P, phi P), up to 136 positions, 8 per word
  uint32_t codes_q[17];  // pair (Q, phi Q)
  int len;
  uint32_t neg;               // bit 0: k1 < 0, bit 1: k2 < 0, bit 2: l1 < 0, bit 3: l2 < 0
  uint32_t s_lo[8], s_hi[8];  // canonical scalars (slow path only)
};

template <int BF>
__device__ __attribute__((noinline)) Xyzz mul_bits(const Aff& p, const uint32_t* k) {
  using E = Ec<BF>;
  Xyzz acc = E::identity();
  for (int i = 255; i >= 0; i--) {
    acc = E::dbl(acc);
    if ((k[i >> 5] >> (i & 31)) & 1u) E::madd(acc, p);
  }
  return acc;
}

// a + b and a - b in affine for two finite points with different x, given inv = 1 / (b.x - a.x)
template <int BF>
__device__ __forceinline__ void affine_sum_diff(const Aff& a, const Aff& b, const Fe& inv, Aff& sum, Aff& dif) {
  using F = Fp<BF>;
  Fe lam = F::mul(F::sub(b.y, a.y), inv);
  Fe x3 = F::carry(F::sub(F::sub(F::sqr(lam), a.x), b.x));
  sum.x = x3;
  sum.y = F::carry(F::sub(F::mul(lam, F::sub(a.x, x3)), a.y));
  sum.inf = false;
  Fe lam2 = F::mul(F::neg(F::add(b.y, a.y)), inv);
  Fe x4 = F::carry(F::sub(F::sub(F::sqr(lam2), a.x), b.x));
  dif.x = x4;
  dif.y = F::carry(F::sub(F::mul(lam2, F::sub(a.x, x4)), a.y));
  dif.inf = false;
}

template <int BF>
__global__ void __launch_bounds__(256) k_fold_points(const uint32_t* __restrict__ g0, uint32_t* __restrict__ o0,
                                                     const uint32_t* __restrict__ g1, uint32_t* __restrict__ o1,
                                                     const Plans2<FoldPlan> plans, uint32_t n) {
  using F = Fp<BF>;
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* g = blockIdx.y == 0 ? g0 : g1;
  uint32_t* o = blockIdx.y == 0 ? o0 : o1;
  const FoldPlan* f = &plans.p[blockIdx.y];
  Aff P = IO::load_dev_affine(g + 16ull * i);
  Aff Q = IO::load_dev_affine(g + 16ull * (i + n));
  Xyzz acc;
  if (P.inf || Q.inf) {
    // padding (identity points): two independent double-and-add ladders
    Xyzz r1 = mul_bits<BF>(P, f->s_lo);
    Xyzz r2 = mul_bits<BF>(Q, f->s_hi);
    acc = E::add(r1, r2);
  } else {
    const uint32_t neg = f->neg;
    const Fe zeta = F::from_const(CurveConsts<BF>::ZETA261);
    // table: T[0..3] = P1, P2 = phi(P1), P1 + P2, P1 - P2 (signs of k1, k2 applied); T[4..7] same for Q
    Aff T[8];
    T[0] = P;
    T[1].x = F::mul(P.x, zeta);
    T[1].y = P.y;
    T[1].inf = false;
    T[4] = Q;
    T[5].x = F::mul(Q.x, zeta);
    T[5].y = Q.y;
    T[5].inf = false;
    if (neg & 1u) T[0].y = F::carry(F::neg(T[0].y));
    if (neg & 2u) T[1].y = F::carry(F::neg(T[1].y));
    if (neg & 4u) T[4].y = F::carry(F::neg(T[4].y));
    if (neg & 8u) T[5].y = F::carry(F::neg(T[5].y));
    // x never vanishes on a prime-order curve y^2 = x^3 + 5, so zeta x != x: one shared inversion
    Fe dp = F::carry(F::sub(T[1].x, T[0].x)), dq = F::carry(F::sub(T[5].x, T[4].x));
    Fe ti = F::inv(F::mul(dp, dq));
    affine_sum_diff<BF>(T[0], T[1], F::mul(ti, dq), T[2], T[3]);
    affine_sum_diff<BF>(T[4], T[5], F::mul(ti, dp), T[6], T[7]);
    acc = E::identity();
    for (int pos = f->len - 1; pos >= 0; pos--) {
      acc = E::dbl(acc);
      const uint32_t cp = (f->codes_p[pos >> 3] >> ((pos & 7) * 4)) & 15u;
      const uint32_t cq = (f->codes_q[pos >> 3] >> ((pos & 7) * 4)) & 15u;
#pragma unroll 1
      for (int h = 0; h < 2; h++) {
        const uint32_t code = h ? cq : cp;
        const uint32_t sel = code & 7u;
        if (sel) {
          Aff t = T[4 * h + (int)sel - 1];
          if (code & 8u) t = E::neg(t);
          E::madd(acc, t);
        }
      }
    }
  }
  Aff r = E::to_affine(acc);
  IO::store_dev_affine(o + 16ull * i, r);
}

// ---------------------------------------------------------------------------
// Monic generator fold:  out[i] = g[i] + s * g[i+n]   (has_p)   or   out[i] = s * g[i]   (!has_p).
// The prover's session (bpgpu.hip, bpgpu_ipa) writes  x^-1 A_lo + x A_hi = x^-1 (A_lo + x^2 A_hi)  and
// carries the common factor as a scalar, so a fold is ONE shared-scalar multiplication per pair.
//   * s = k1 + k2 lambda (GLV) is an Eisenstein integer k1 + k2 w; it is expanded in base 2 with
//     width 2: the digits are 0, the six units {+-1, +-w, +-(1 + w)} and their products with
//     alpha = 1 - w (the twelve odd residues modulo 4), so that one position in three is non-zero
//     (bpgpu.hip, eisenstein_codes).  The points of the unit digits need no arithmetic:
//     phi Q = (zeta x, y) and, because 1 + lambda + lambda^2 = 0,  Q + phi Q = -phi^2 Q = (zeta^2 x, -y);
//     those of the other six are the same maps applied to Q' = alpha Q = Q - phi(Q), ONE affine addition
//     per generator whose inversion (of (zeta - 1) x) is shared by the workgroup.  Q' lives in LDS.
//   * the inversion back to affine is shared by the 256 lanes of the workgroup through a product tree in LDS.
// 4-bit codes, position 0 = least significant:  code & 3: 0 none, 1: x, 2: zeta x, 3: zeta^2 x;
// code & 4: of Q' instead of Q;  code & 8: use -y.
// ---------------------------------------------------------------------------
struct MonicPlan {
  uint32_t codes[17];  // up to 136 positions, 8 per word
  int len;
};

// 1 / v for every lane of a 256-thread workgroup (v != 0 in every lane); all lanes must call.
template <int BF>
__device__ __forceinline__ Fe block_inverse_256(const Fe& v, uint32_t (*tree)[9]) {
  using F = Fp<BF>;
  const uint32_t tid = threadIdx.x;
  auto ld = [&](uint32_t node) {
    Fe r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = (int32_t)tree[node][k];
    return r;
  };
  auto st = [&](uint32_t node, const Fe& a) {
#pragma unroll
    for (int k = 0; k < 9; k++) tree[node][k] = (uint32_t)a.l[k];
  };
  st(256 + tid, F::carry(v));
  __syncthreads();
  for (uint32_t w = 128; w >= 1; w >>= 1) {
    if (tid < w) st(w + tid, F::mul(ld(2 * (w + tid)), ld(2 * (w + tid) + 1)));
    __syncthreads();
  }
  // the root is inverted by one lane; which wave does it rotates with the workgroup so that
  // co-resident workgroups do not queue their inversions on one SIMD
  if (tid == 64 * (blockIdx.x & 3u)) st(1, F::inv(ld(1)));
  __syncthreads();
  for (uint32_t w = 1; w <= 128; w <<= 1) {
    if (tid < w) {
      const uint32_t p = w + tid;
      Fe pi = ld(p), l = ld(2 * p), r = ld(2 * p + 1);
      st(2 * p, F::mul(pi, r));
      st(2 * p + 1, F::mul(pi, l));
    }
    __syncthreads();
  }
  return ld(256 + tid);
}

// (zeta - 1) x of a finite point: the denominator of the slope of Q - phi(Q); never zero (x != 0, zeta != 1)
template <int BF>
__device__ __forceinline__ Fe alpha_den(const Aff& Q, const Fe& zeta) {
  using F = Fp<BF>;
  return F::carry(F::sub(F::mul(Q.x, zeta), Q.x));
}
// Q' = Q - phi(Q) = (x, y) + (zeta x, -y) for a finite Q, given inv = 1 / ((zeta - 1) x):
// slope = -2y inv,  x' = slope^2 - (1 + zeta) x = slope^2 + zeta^2 x,  y' = slope (x - x') - y
template <int BF>
__device__ __forceinline__ Aff alpha_point(const Aff& Q, const Fe& inv, const Fe& zeta2) {
  using F = Fp<BF>;
  Aff r;
  const Fe lam = F::mul(F::neg(F::add(Q.y, Q.y)), inv);
  r.x = F::carry(F::add(F::sqr(lam), F::mul(Q.x, zeta2)));
  r.y = F::carry(F::sub(F::mul(lam, F::sub(Q.x, r.x)), Q.y));
  r.inf = false;
  return r;
}

// ---- GLV split, shared by the host (shared fold multipliers, bpgpu.hip) and the device (k_hadamard_points) ----
// 256-bit wrapping helpers
BP_HD void mul256_wrap(const uint64_t (&a)[4], const uint64_t (&b)[4], uint64_t (&out)[4]) {
  uint64_t r[4] = {0, 0, 0, 0};
  BP_UNROLL
  for (int i = 0; i < 4; i++) {
    unsigned __int128 carry = 0;
    BP_UNROLL
    for (int j = 0; j < 4; j++) {
      if (i + j < 4) {
        carry += (unsigned __int128)a[i] * b[j] + r[i + j];
        r[i + j] = (uint64_t)carry;
        carry >>= 64;
      }
    }
  }
  BP_UNROLL
  for (int i = 0; i < 4; i++) out[i] = r[i];
}
BP_HD void sub256_wrap(uint64_t (&a)[4], const uint64_t (&b)[4]) {
  unsigned __int128 br = 0;
  BP_UNROLL
  for (int i = 0; i < 4; i++) {
    unsigned __int128 d = (unsigned __int128)a[i] - b[i] - br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
BP_HD void neg256_wrap(uint64_t (&a)[4]) {
  uint64_t z[4] = {0, 0, 0, 0};
  sub256_wrap(z, a);
  BP_UNROLL
  for (int i = 0; i < 4; i++) a[i] = z[i];
}
// (k * g) >> 384 for a 4-limb k and a 5-limb g
BP_HD void mul_shift384(const uint64_t (&k)[4], const uint64_t (&g)[5], uint64_t (&out)[4]) {
  uint64_t r[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  BP_UNROLL
  for (int i = 0; i < 4; i++) {
    unsigned __int128 carry = 0;
    BP_UNROLL
    for (int j = 0; j < 5; j++) {
      carry += (unsigned __int128)k[i] * g[j] + r[i + j];
      r[i + j] = (uint64_t)carry;
      carry >>= 64;
    }
    r[i + 5] = (uint64_t)carry;
  }
  out[0] = r[6];
  out[1] = r[7];
  out[2] = r[8];
  out[3] = 0;
}
// k (canonical, < r) = k1 + k2 lambda (mod r) with short k1, k2 (below 2^129); magnitudes and signs out.
template <int CURVE>
BP_HD void glv_split(const uint64_t (&k)[4], uint64_t (&k1)[4], bool& neg1, uint64_t (&k2)[4], bool& neg2) {
  using CC = CurveConsts<CURVE>;
  uint64_t c1[4], c2[4], t[4];
  mul_shift384(k, CC::GLV_G1, c1);
  mul_shift384(k, CC::GLV_G2, c2);
  if (CC::GLV_N1_NEG) neg256_wrap(c1);
  if (CC::GLV_N2_NEG) neg256_wrap(c2);
  BP_UNROLL
  for (int i = 0; i < 4; i++) {
    k1[i] = k[i];
    k2[i] = 0;
  }
  mul256_wrap(c1, CC::GLV_A1, t);
  sub256_wrap(k1, t);
  mul256_wrap(c2, CC::GLV_A2, t);
  sub256_wrap(k1, t);
  mul256_wrap(c1, CC::GLV_B1, t);
  sub256_wrap(k2, t);
  mul256_wrap(c2, CC::GLV_B2, t);
  sub256_wrap(k2, t);
  neg1 = (k1[3] >> 63) != 0;
  neg2 = (k2[3] >> 63) != 0;
  if (neg1) neg256_wrap(k1);
  if (neg2) neg256_wrap(k2);
}

// Vector y (blockIdx.y = 0 / 1):  o[i] = p[i] + s q[i]  (has_p)  or  s q[i]  (!has_p; p is not read).
// o may alias p or q: a lane reads its own elements before it writes.
template <int BF>
__global__ void __launch_bounds__(256) k_fold_monic(const uint32_t* p0, const uint32_t* q0, uint32_t* o0,
                                                    const uint32_t* p1, const uint32_t* q1, uint32_t* o1,
                                                    const Plans2<MonicPlan> plans, uint32_t n, int has_p) {
  using F = Fp<BF>;
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  __shared__ uint32_t tree[512][9];
  __shared__ int32_t qa[45][256];  // Q' of this lane: x', zeta x', zeta^2 x', y', -y' (9 limbs each)
  const uint32_t tid = threadIdx.x;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  const uint32_t* gp = blockIdx.y == 0 ? p0 : p1;
  const uint32_t* gq = blockIdx.y == 0 ? q0 : q1;
  uint32_t* o = blockIdx.y == 0 ? o0 : o1;
  const MonicPlan* f = &plans.p[blockIdx.y];
  const Fe zeta = F::from_const(CurveConsts<BF>::ZETA261);
  Aff Q;
  Q.inf = true;
  if (active) Q = IO::load_dev_affine(gq + 16ull * i);
  const bool qfin = active && !Q.inf;
  // x never vanishes on a prime-order curve y^2 = x^3 + 5, so Q, phi Q, phi^2 Q are distinct
  const Fe ainv = block_inverse_256<BF>(qfin ? alpha_den<BF>(Q, zeta) : F::one(), tree);
  Jac acc = E::jidentity();  // Jacobian: the chain is mostly doublings
  if (qfin) {
    const Fe zeta2 = F::sqr(zeta);
    const Fe x2 = F::mul(Q.x, zeta), x3 = F::mul(Q.x, zeta2);
    const Fe yn = F::carry(F::neg(Q.y));
    {
      const Aff A = alpha_point<BF>(Q, ainv, zeta2);
      const Fe ax2 = F::mul(A.x, zeta), ax3 = F::mul(A.x, zeta2), ayn = F::carry(F::neg(A.y));
#pragma unroll
      for (int k = 0; k < 9; k++) {
        qa[k][tid] = A.x.l[k];
        qa[9 + k][tid] = ax2.l[k];
        qa[18 + k][tid] = ax3.l[k];
        qa[27 + k][tid] = A.y.l[k];
        qa[36 + k][tid] = ayn.l[k];
      }
    }
    for (int pos = f->len - 1; pos >= 0; pos--) {
      acc = E::jdbl(acc);
      const uint32_t code = (f->codes[pos >> 3] >> ((pos & 7) * 4)) & 15u;  // wave-uniform
      const uint32_t sel = code & 3u;
      if (sel) {
        Aff t;
        t.inf = false;
        if (code & 4u) {  // a digit of the alpha orbit: the lane's own column of qa (no other lane touches it)
          const int xo = 9 * ((int)sel - 1), yo = (code & 8u) ? 36 : 27;
#pragma unroll
          for (int k = 0; k < 9; k++) {
            t.x.l[k] = qa[xo + k][tid];
            t.y.l[k] = qa[yo + k][tid];
          }
        } else {
          t.x = sel == 1 ? Q.x : (sel == 2 ? x2 : x3);
          t.y = (code & 8u) ? yn : Q.y;
        }
        E::jmadd(acc, t);
      }
    }
  }
  if (active && has_p) E::jmadd(acc, IO::load_dev_affine(gp + 16ull * i));
  // back to affine: x = X / Z^2, y = Y / Z^3 with one shared inversion of Z
  const bool fin = active && !acc.inf;
  const Fe inv = block_inverse_256<BF>(fin ? acc.z : F::one(), tree);
  if (active) IO::store_dev_affine(o + 16ull * i, E::jto_affine_with(acc, inv));
}

// ---------------------------------------------------------------------------
// Up to MULTI_MAX_M shared multipliers on ONE doubling chain:
//   o[i] = [p[i] +] s_0 q[i] + s_1 q[n + i] + ... + s_{m-1} q[(m-1) n + i],   i < n
// -- the k-ary fold sum_vecs(chunks.zip(y).map(scale_vec)) of src/proofs/bp_rec_kary.rs:105-112 ==
// src/reductions/ipa_to_bp_unroll.rs:253-260 in one pass (k - 1 <= 7 multipliers and chunk 0 as p when y[0] = 1).
// Horner's rule over k_fold_monic walks k - 1 chains of ~128 doublings + ~55 additions one after the other; here
// one chain of ~128 doublings carries all the additions (7 multipliers: 128 + 385 group operations instead of 1281),
// and where the launch is a wave per SIMD (ck <= 2^16) the time follows the chain's length.
// The operands do not fit registers or LDS (7 generators and their alpha-multiples are 28 field elements a lane), so
// they stay in GLOBAL memory: the kernel first writes Q'_t = alpha Q_t (one affine addition each, ONE workgroup-shared
// inversion for all of a lane's generators) to `alpha`, then walks a host-made list of operations -- "d doublings,
// then add the point of digit code c of generator t" -- and loads the operand of the NEXT operation while the
// current one runs (a lane reads only its own column: coalesced 64-byte rows; 2 ms of arithmetic hide the latency).
// o may alias p, and q as well (the two-multiplier fold in place): a lane reads its own column only and the output's
// one store comes after every read of it; alpha must not overlap either.
// ---------------------------------------------------------------------------
constexpr int MULTI_MAX_M = 7;
constexpr int MULTI_MAX_OPS = 768;
// op: bits 0-3 digit code (as in MonicPlan), 4-6 generator t, 7-11 doublings before the addition, 12 an addition follows.
// Short vectors (fewer outputs than the chip has lanes at a wave per SIMD): `lanes` = 2 or 4 lanes share an output, each
// with its own operation list op[start[r] .. start[r + 1]) -- every list doubles through all positions, the additions
// are dealt out in turn -- and the partial sums meet in LDS (as in k_fold_joint4_split).  The time of such a launch is its
// longest chain: 128 doublings + 381 / lanes additions.
struct MultiOps {
  uint16_t op[MULTI_MAX_OPS];
  uint32_t start[5];
  uint32_t lanes;  // 1, 2 or 4
  uint32_t m;
};
template <int BF>
__global__ void __launch_bounds__(256) k_fold_multi(const uint32_t* q, const uint32_t* p, uint32_t* __restrict__ alpha,
                                                    uint32_t* o, const MultiOps ops, uint32_t n) {
  using F = Fp<BF>;
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  __shared__ uint32_t tree[512][9];
  __shared__ int32_t xch[3 * 37 * 64];  // XYZZ partial sums of roles 1 .. lanes-1 (36 limbs + identity flag per output)
  const uint32_t lanes = ops.lanes, per_wg = 256u / lanes;          // wave-uniform
  const uint32_t col = threadIdx.x % per_wg, role = threadIdx.x / per_wg;  // a wave has one role
  const uint32_t i = blockIdx.x * per_wg + col;
  const bool active = i < n;
  const int m = (int)ops.m;
  const Fe zeta = F::from_const(CurveConsts<BF>::ZETA261);
  const Fe zeta2 = F::sqr(zeta);
  {
    // Q'_t = Q_t - phi(Q_t) for the generators t = role (mod lanes) of the lane's output: prefix products of the slopes'
    // denominators, one inversion
    Fe pre[MULTI_MAX_M];
    Fe run = F::one();
#pragma unroll
    for (int t = 0; t < MULTI_MAX_M; t++) {
      pre[t] = run;
      if (t < m && active && (uint32_t)t % lanes == role) {
        const Aff Q = IO::load_dev_affine(q + 16ull * ((uint64_t)t * n + i));
        if (!Q.inf) run = F::mul(run, alpha_den<BF>(Q, zeta));
      }
    }
    Fe inv = block_inverse_256<BF>(run, tree);
#pragma unroll
    for (int t = MULTI_MAX_M - 1; t >= 0; t--) {
      if (t < m && active && (uint32_t)t % lanes == role) {
        const Aff Q = IO::load_dev_affine(q + 16ull * ((uint64_t)t * n + i));
        Aff A;
        A.inf = true;
        A.x = A.y = F::zero();
        if (!Q.inf) {
          const Fe it = F::mul(inv, pre[t]);
          inv = F::mul(inv, alpha_den<BF>(Q, zeta));
          A = alpha_point<BF>(Q, it, zeta2);
        }
        IO::store_dev_affine(alpha + 16ull * ((uint64_t)t * n + i), A);  // read back by the lanes of this output only
      }
    }
    if (lanes > 1) {  // the other roles of the output (same workgroup) read these rows
      __threadfence();
      __syncthreads();
    }
  }
  Jac acc = E::jidentity();
  const uint32_t j0 = ops.start[role], j1 = ops.start[role + 1];
  if (active && j0 < j1) {
    auto operand = [&](uint32_t op) -> const uint4* {
      const uint32_t t = (op >> 4) & 7u;
      return reinterpret_cast<const uint4*>(((op & 4u) ? alpha : q) + 16ull * ((uint64_t)t * n + i));
    };
    const uint4* pp = operand(ops.op[j0]);
    uint4 r0 = pp[0], r1 = pp[1], r2 = pp[2], r3 = pp[3];
    for (uint32_t j = j0; j < j1; j++) {
      const uint32_t op = ops.op[j];  // wave-uniform
      pp = operand(ops.op[j + 1 < j1 ? j + 1 : j]);
      const uint4 n0 = pp[0], n1 = pp[1], n2 = pp[2], n3 = pp[3];
      for (uint32_t d = (op >> 7) & 31u; d > 0; d--) acc = E::jdbl(acc);
      if (op & 0x1000u) {
        uint32_t w[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
        Aff pt = IO::load_dev_affine(w);
        const uint32_t sel = op & 3u;
        if (sel != 1) pt.x = F::mul(pt.x, sel == 2 ? zeta : zeta2);
        if (op & 8u) pt.y = F::carry(F::neg(pt.y));
        E::jmadd(acc, pt);  // skips an identity operand (zero padding)
      }
      r0 = n0;
      r1 = n1;
      r2 = n2;
      r3 = n3;
    }
  }
  if (active && p && role == 0) E::jmadd(acc, IO::load_dev_affine(p + 16ull * i));
  if (lanes == 1) {  // one lane an output: straight to affine through the shared inversion of Z
    const bool fin = active && !acc.inf;
    const Fe inv = block_inverse_256<BF>(fin ? acc.z : F::one(), tree);
    if (active) IO::store_dev_affine(o + 16ull * i, E::jto_affine_with(acc, inv));
    return;
  }
  // several lanes an output: the partial sums meet in LDS as XYZZ (Ec::add is the tested general addition)
  Xyzz r = E::identity();
  if (active && !acc.inf) {
    r.x = acc.x;
    r.y = acc.y;
    r.zz = F::sqr(acc.z);
    r.zzz = F::mul(r.zz, acc.z);
    r.inf = false;
  }
  if (role != 0) {
    int32_t* x = xch + (size_t)(role - 1) * 37 * per_wg;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      x[k * per_wg + col] = r.x.l[k];
      x[(9 + k) * per_wg + col] = r.y.l[k];
      x[(18 + k) * per_wg + col] = r.zz.l[k];
      x[(27 + k) * per_wg + col] = r.zzz.l[k];
    }
    x[36 * per_wg + col] = r.inf ? 1 : 0;
  }
  __syncthreads();
  if (role == 0) {
    for (uint32_t rr = 1; rr < lanes; rr++) {
      const int32_t* x = xch + (size_t)(rr - 1) * 37 * per_wg;
      Xyzz b;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        b.x.l[k] = x[k * per_wg + col];
        b.y.l[k] = x[(9 + k) * per_wg + col];
        b.zz.l[k] = x[(18 + k) * per_wg + col];
        b.zzz.l[k] = x[(27 + k) * per_wg + col];
      }
      b.inf = x[36 * per_wg + col] != 0;
      r = E::add(r, b);
    }
  }
  const bool fin = role == 0 && active && !r.inf;
  const Fe inv = block_inverse_256<BF>(fin ? F::mul(r.zz, r.zzz) : F::one(), tree);
  if (role == 0 && active) {
    Aff a;
    a.inf = !fin;
    a.x = F::zero();
    a.y = F::zero();
    if (fin) {
      a.x = F::mul(r.x, F::mul(inv, r.zzz));
      a.y = F::mul(r.y, F::mul(inv, r.zz));
    }
    IO::store_dev_affine(o + 16ull * i, a);
  }
}

// ---------------------------------------------------------------------------
// Two fold rounds in one pass:  out[i] = g[i] + s1 g[i+q] + s2 g[i+2q] + s3 g[i+3q],  i < q.
// The prover (bpgpu.hip, "pair folding") skips the generator fold of every other long round -- that
// round's L and R are MSMs over the unfolded generators with weighted scalars, as in the deferred
// mode -- and then applies both challenges at once:
//   x1^-1 (x0^-1 A_i + x0 A_{i+2q}) + x1 (x0^-1 A_{i+q} + x0 A_{i+3q})
//     = (x0 x1)^-1 [ A_i + x1^2 A_{i+q} + x0^2 A_{i+2q} + x0^2 x1^2 A_{i+3q} ].
// The three shared multipliers ride ONE doubling chain (128 doublings for three multiplications
// instead of 3 x 128 when the rounds fold one after the other); digits are Eisenstein units as in
// k_fold_monic; zeta x of the three generators is kept in registers, a unit's x of the Q' orbit is formed on the
// fly (one multiplication by zeta or zeta^2).  out may alias g: a lane reads only indices = i (mod q).
// ---------------------------------------------------------------------------
struct Joint4Plan {
  MonicPlan m[3];  // multipliers of g[i+q], g[i+2q], g[i+3q]
  int len;         // positions of the longest expansion
};
template <int BF>
__global__ void __launch_bounds__(256) k_fold_joint4(const uint32_t* g0, uint32_t* o0, const uint32_t* g1, uint32_t* o1,
                                                     const Plans2<Joint4Plan> plans, uint32_t q) {
  using F = Fp<BF>;
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  __shared__ uint32_t tree[512][9];
  __shared__ int32_t qa[54][256];  // Q'_t = alpha Q_t of this lane, t = 0..2: x' (9 limbs), y' (9 limbs)
  const uint32_t tid = threadIdx.x;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < q;
  const uint32_t* g = blockIdx.y == 0 ? g0 : g1;
  uint32_t* o = blockIdx.y == 0 ? o0 : o1;
  const Joint4Plan* f = &plans.p[blockIdx.y];
  const Fe zeta = F::from_const(CurveConsts<BF>::ZETA261);
  const Fe zeta2 = F::sqr(zeta);
  Aff Q0, Q1, Q2;
  Q0.inf = Q1.inf = Q2.inf = true;
  if (active) {
    Q0 = IO::load_dev_affine(g + 16ull * (i + q));
    Q1 = IO::load_dev_affine(g + 16ull * (i + 2ull * q));
    Q2 = IO::load_dev_affine(g + 16ull * (i + 3ull * q));
  }
  {
    // Q'_t = Q_t - phi(Q_t): the three slopes share one inversion (and that one the workgroup's)
    const Fe d0 = Q0.inf ? F::one() : alpha_den<BF>(Q0, zeta);
    const Fe d1 = Q1.inf ? F::one() : alpha_den<BF>(Q1, zeta);
    const Fe d2 = Q2.inf ? F::one() : alpha_den<BF>(Q2, zeta);
    const Fe d01 = F::mul(d0, d1);
    const Fe iv = block_inverse_256<BF>(F::mul(d01, d2), tree);
    const Fe i2 = F::mul(iv, d01), i01 = F::mul(iv, d2);
    const Fe i0 = F::mul(i01, d1), i1 = F::mul(i01, d0);
#pragma unroll 1
    for (int t = 0; t < 3; t++) {
      const Aff& Qt = t == 0 ? Q0 : (t == 1 ? Q1 : Q2);
      const Fe& it = t == 0 ? i0 : (t == 1 ? i1 : i2);
      if (!Qt.inf) {
        const Aff A = alpha_point<BF>(Qt, it, zeta2);
#pragma unroll
        for (int k = 0; k < 9; k++) {
          qa[18 * t + k][tid] = A.x.l[k];
          qa[18 * t + 9 + k][tid] = A.y.l[k];
        }
      }
    }
  }
  Jac acc = E::jidentity();  // Jacobian: the chain is mostly doublings
  if (active) {
    // zeta x of the three generators stays in registers: their unit digits then need no multiplication
    // (zeta^2 x = -(x + zeta x), since 1 + zeta + zeta^2 = 0); the digits of Q' (in LDS) still form theirs on the fly
    const Fe zx0 = F::mul(Q0.x, zeta), zx1 = F::mul(Q1.x, zeta), zx2 = F::mul(Q2.x, zeta);
    for (int pos = f->len - 1; pos >= 0; pos--) {
      acc = E::jdbl(acc);
#pragma unroll 1
      for (int t = 0; t < 3; t++) {
        const uint32_t code = (f->m[t].codes[pos >> 3] >> ((pos & 7) * 4)) & 15u;  // wave-uniform
        const uint32_t sel = code & 3u;
        if (sel) {
          Aff p;  // uniform selects, not an indexed array (which would live in scratch)
          p.inf = t == 0 ? Q0.inf : (t == 1 ? Q1.inf : Q2.inf);
          if (code & 4u) {  // a digit of the alpha orbit: the lane's own column of qa
#pragma unroll
            for (int k = 0; k < 9; k++) {
              p.x.l[k] = qa[18 * t + k][tid];
              p.y.l[k] = qa[18 * t + 9 + k][tid];
            }
            if (sel != 1) p.x = F::mul(p.x, sel == 2 ? zeta : zeta2);
          } else {
            p.x = t == 0 ? Q0.x : (t == 1 ? Q1.x : Q2.x);
            p.y = t == 0 ? Q0.y : (t == 1 ? Q1.y : Q2.y);
            if (sel != 1) {
              const Fe zx = t == 0 ? zx0 : (t == 1 ? zx1 : zx2);
              p.x = sel == 2 ? zx : F::carry(F::neg(F::add(p.x, zx)));
            }
          }
          if (code & 8u) p.y = F::carry(F::neg(p.y));
          E::jmadd(acc, p);  // skips an identity operand (zero padding)
        }
      }
    }
    E::jmadd(acc, IO::load_dev_affine(g + 16ull * i));
  }
  const bool fin = active && !acc.inf;
  const Fe inv = block_inverse_256<BF>(fin ? acc.z : F::one(), tree);
  if (active) IO::store_dev_affine(o + 16ull * i, E::jto_affine_with(acc, inv));
}

// The same fold for short vectors (q <= 2^14: fewer outputs than the chip has SIMDs x 64 lanes, so the kernel above
// is one long chain of 128 doublings and 165 additions per lane with half the SIMDs idle, and a lone wave runs no
// slower than one that shares its SIMD).  Here two lanes share an output: both walk the doubling chain, lane 0 with
// the digits of s1 and the even positions of s2 (and P at the end), lane 1 with those of s3 and the odd positions of
// s2 -- any split of the (position, multiplier) pairs is a valid split of the sum, and this one is uniform per wave
// (waves 0-1 of a workgroup are lane 0 of 128 outputs, waves 2-3 lane 1).  83 additions a lane instead of 165; the
// two halves meet in LDS (as XYZZ: Ec::add is the tested general addition) and lane 0 normalises through the
// workgroup-shared inversion.  Twice the doublings in all, 0.7 of the latency where latency is all there is.
// (Three lanes an output, one per multiplier, gain nothing at q = 2^14: 1536 waves on 1024 SIMDs.)
template <int BF>
__global__ void __launch_bounds__(256) k_fold_joint4_split(const uint32_t* g0, uint32_t* o0, const uint32_t* g1, uint32_t* o1,
                                                           const Plans2<Joint4Plan> plans, uint32_t q) {
  using F = Fp<BF>;
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  __shared__ uint32_t tree[512][9];
  __shared__ int32_t xch[37][128];  // XYZZ results of the second lanes (36 limbs + identity flag)
  const uint32_t tid = threadIdx.x, col = tid & 127u, role = tid >> 7;
  const uint32_t i = blockIdx.x * 128 + col;
  const bool active = i < q;
  const uint32_t* g = blockIdx.y == 0 ? g0 : g1;
  uint32_t* o = blockIdx.y == 0 ? o0 : o1;
  const Joint4Plan* f = &plans.p[blockIdx.y];
  const Fe zeta = F::from_const(CurveConsts<BF>::ZETA261);
  // generator a: all digits of multiplier ta; generator b = Q_2 (multiplier index 1): the positions of parity `role`
  const int ta = role == 0 ? 0 : 2;
  Aff Qa, Qb;
  Qa.inf = Qb.inf = true;
  if (active) {
    Qa = IO::load_dev_affine(g + 16ull * (i + (uint64_t)(ta + 1) * q));
    Qb = IO::load_dev_affine(g + 16ull * (i + 2ull * q));
  }
  Aff Aa, Ab;  // alpha Q
  Aa.inf = Qa.inf;
  Ab.inf = Qb.inf;
  {
    const Fe da = Qa.inf ? F::one() : alpha_den<BF>(Qa, zeta), db = Qb.inf ? F::one() : alpha_den<BF>(Qb, zeta);
    const Fe iv = block_inverse_256<BF>(F::mul(da, db), tree);
    const Fe zeta2 = F::sqr(zeta);
    Aa = Qa.inf ? Qa : alpha_point<BF>(Qa, F::mul(iv, db), zeta2);
    Ab = Qb.inf ? Qb : alpha_point<BF>(Qb, F::mul(iv, da), zeta2);
  }
  Jac acc = E::jidentity();
  if (active) {
    const Fe zxa = F::mul(Qa.x, zeta), zaa = F::mul(Aa.x, zeta), zxb = F::mul(Qb.x, zeta), zab = F::mul(Ab.x, zeta);
    for (int pos = f->len - 1; pos >= 0; pos--) {
      acc = E::jdbl(acc);
#pragma unroll 1
      for (int h = 0; h < 2; h++) {  // h = 0: generator a, h = 1: generator b
        if (h == 1 && (uint32_t)(pos & 1) != role) continue;  // wave-uniform
        const MonicPlan* m = &f->m[h == 0 ? ta : 1];
        const uint32_t code = pos < m->len ? (m->codes[pos >> 3] >> ((pos & 7) * 4)) & 15u : 0u;
        const uint32_t sel = code & 3u;
        if (!sel) continue;
        const bool alt = (code & 4u) != 0;  // a digit of the alpha orbit
        Aff p;
        p.inf = h == 0 ? Qa.inf : Qb.inf;
        p.x = h == 0 ? (alt ? Aa.x : Qa.x) : (alt ? Ab.x : Qb.x);
        p.y = h == 0 ? (alt ? Aa.y : Qa.y) : (alt ? Ab.y : Qb.y);
        if (sel != 1) {
          const Fe z1 = h == 0 ? (alt ? zaa : zxa) : (alt ? zab : zxb);
          p.x = sel == 2 ? z1 : F::carry(F::neg(F::add(p.x, z1)));  // zeta^2 x = -(x + zeta x)
        }
        if (code & 8u) p.y = F::carry(F::neg(p.y));
        E::jmadd(acc, p);  // skips an identity operand (zero padding)
      }
    }
    if (role == 0) E::jmadd(acc, IO::load_dev_affine(g + 16ull * i));
  }
  Xyzz r = E::identity();
  if (active && !acc.inf) {
    r.x = acc.x;
    r.y = acc.y;
    r.zz = F::sqr(acc.z);
    r.zzz = F::mul(r.zz, acc.z);
    r.inf = false;
  }
  if (role == 1) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
      xch[k][col] = r.x.l[k];
      xch[9 + k][col] = r.y.l[k];
      xch[18 + k][col] = r.zz.l[k];
      xch[27 + k][col] = r.zzz.l[k];
    }
    xch[36][col] = r.inf ? 1 : 0;
  }
  __syncthreads();
  if (role == 0) {
    Xyzz b;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      b.x.l[k] = xch[k][col];
      b.y.l[k] = xch[9 + k][col];
      b.zz.l[k] = xch[18 + k][col];
      b.zzz.l[k] = xch[27 + k][col];
    }
    b.inf = xch[36][col] != 0;
    r = E::add(r, b);
  }
  const bool fin = role == 0 && active && !r.inf;
  const Fe inv = block_inverse_256<BF>(fin ? F::mul(r.zz, r.zzz) : F::one(), tree);
  if (role == 0 && active) {
    Aff a;
    a.inf = !fin;
    a.x = F::zero();
    a.y = F::zero();
    if (fin) {
      a.x = F::mul(r.x, F::mul(inv, r.zzz));
      a.y = F::mul(r.y, F::mul(inv, r.zz));
    }
    IO::store_dev_affine(o + 16ull * i, a);
  }
}

// ---------------------------------------------------------------------------
// Fixed-base table of the prover's frozen generators (MsmPlan::table_stride).  The short rounds of a proof run
// fifteen pairs of MSMs over the SAME points; with the multiples 2^(c w) P at hand every window's digits share
// one bucket set and the host tail has nothing to double.
//   k_table_chain: one lane per point (A0 | B0 | Q, `stride` = 2 n0 + 1 of them): copies P to window 0 and
//                  walks the doubling chain (Jacobian: cheapest doubling), leaving 2^(c w) P for w = 1 .. W-1 in tmp;
//   k_table_norm:  lanes over the (w, point) elements: Jacobian -> affine with the workgroup-shared inversion.
// ---------------------------------------------------------------------------
template <int BF>
__global__ void __launch_bounds__(256) k_table_chain(const uint32_t* __restrict__ pts, uint32_t np, uint32_t n0,
                                                     uint32_t* __restrict__ table, uint32_t* __restrict__ tmp, int c, int W) {
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  const uint32_t stride = 2 * n0 + 1;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= stride) return;
  const uint64_t src = j < n0 ? j : (j < 2 * n0 ? (uint64_t)np + (j - n0) : 2ull * np);
  const Aff P = IO::load_dev_affine(pts + 16ull * src);
  IO::store_dev_affine(table + 16ull * j, P);
  Jac acc = E::jfrom_affine(P);
  for (int w = 1; w < W; w++) {
    for (int k = 0; k < c; k++) acc = E::jdbl(acc);
    IO::store_dev_jac(tmp + 24ull * ((uint64_t)(w - 1) * stride + j), acc);
  }
}

// A lane converts TABLE_NORM_BATCH elements (e, e + quarter, ...) with one inversion of its own -- the product of
// their denominators -- and that one is shared by the workgroup: four times fewer workgroup inversions, whose
// single-lane root inversion is what this kernel waits for.
constexpr int TABLE_NORM_BATCH = 4;
template <int BF>
__global__ void __launch_bounds__(256) k_table_norm(const uint32_t* __restrict__ tmp, uint32_t* __restrict__ table,
                                                    uint32_t stride, uint32_t count) {
  using F = Fp<BF>;
  using IO = PointIo<BF>;
  __shared__ uint32_t tree[512][9];
  const uint32_t quarter = (count + TABLE_NORM_BATCH - 1) / TABLE_NORM_BATCH;
  const uint32_t e0 = blockIdx.x * blockDim.x + threadIdx.x;
  // prefix products of the denominators Z (1 for identity / absent elements)
  using E = Ec<BF>;
  Fe pre[TABLE_NORM_BATCH];
  Fe run = F::one();
#pragma unroll
  for (int k = 0; k < TABLE_NORM_BATCH; k++) {
    const uint64_t e = (uint64_t)e0 + (uint64_t)k * quarter;
    pre[k] = run;
    if (e0 < quarter && e < count) {
      const Jac p = IO::load_dev_jac(tmp + 24ull * e);
      if (!p.inf) run = F::mul(run, p.z);
    }
  }
  Fe inv = block_inverse_256<BF>(run, tree);  // 1 / (product of this lane's denominators)
#pragma unroll
  for (int k = TABLE_NORM_BATCH - 1; k >= 0; k--) {
    const uint64_t e = (uint64_t)e0 + (uint64_t)k * quarter;
    if (e0 < quarter && e < count) {
      const Jac p = IO::load_dev_jac(tmp + 24ull * e);
      Fe zi = F::one();
      if (!p.inf) {
        zi = F::mul(inv, pre[k]);   // 1 / Z
        inv = F::mul(inv, p.z);     // drop this element from the running inverse
      }
      IO::store_dev_affine(table + 16ull * ((uint64_t)stride + e), E::jto_affine_with(p, zi));
    }
  }
}

// out[i] = s v[i] for two vectors in one launch (blockIdx.y: v0 with s.lo, v1 with s.hi)
template <int SF>
__global__ void __launch_bounds__(256) k_scale_scalars(const uint32_t* __restrict__ v0, uint32_t* __restrict__ o0,
                                                       const uint32_t* __restrict__ v1, uint32_t* __restrict__ o1,
                                                       FoldScalars s, uint32_t n) {
  using F = Fp<SF>;
  const uint32_t* v = blockIdx.y == 0 ? v0 : v1;
  uint32_t* o = blockIdx.y == 0 ? o0 : o1;
  const Fe m = blockIdx.y == 0 ? F::load_ark(s.lo) : F::load_ark(s.hi);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint4* pv = reinterpret_cast<const uint4*>(v + 8ull * i);
    uint4 a0 = pv[0], a1 = pv[1];
    uint32_t w[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    F::store_packed(F::mul(F::unpack(w), m), w);
    uint4* po = reinterpret_cast<uint4*>(o + 8ull * i);
    po[0] = make_uint4(w[0], w[1], w[2], w[3]);
    po[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// ---------------------------------------------------------------------------
// Vector utilities of the k-ary protocols: the generalised fold
// sum_vecs(chunks.zip(y).map(scale_vec)) of bp_rec_kary.rs:100-112 == ipa_to_bp_unroll.rs:248-260,
// hadamard / hadamard_gp / powers of src/util/mod.rs:57-85.  Scalars are in ark form.
// ---------------------------------------------------------------------------
// o[i] = sum_{j<k} y[j] v[j ck + i], i < ck.  y: k multipliers in device memory; their device form
// is staged in LDS once per workgroup (dynamic LDS: 36 k bytes).  o may alias v.
template <int SF>
__global__ void __launch_bounds__(256) k_fold_scalars_kary(const uint32_t* v, const uint32_t* __restrict__ y, uint32_t* o,
                                                           uint32_t ck, uint32_t k) {
  using F = Fp<SF>;
  extern __shared__ int32_t ysh[];
  for (uint32_t j = threadIdx.x; j < k; j += blockDim.x) {
    uint32_t w[8];
#pragma unroll
    for (int t = 0; t < 8; t++) w[t] = y[8ull * j + t];
    const Fe m = F::load_ark(w);
#pragma unroll
    for (int t = 0; t < 9; t++) ysh[9 * j + t] = m.l[t];
  }
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ck; i += gridDim.x * blockDim.x) {
    Fe acc = F::zero();
    for (uint32_t j = 0; j < k; j++) {
      const uint4* pv = reinterpret_cast<const uint4*>(v + 8ull * ((uint64_t)j * ck + i));
      uint4 a0 = pv[0], a1 = pv[1];
      uint32_t w[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      Fe m;
#pragma unroll
      for (int t = 0; t < 9; t++) m.l[t] = ysh[9 * j + t];
      acc = fe_partial_reduce<SF>(F::add(acc, F::mul(F::unpack(w), m)));
    }
    uint32_t w[8];
    F::store_packed(acc, w);
    uint4* po = reinterpret_cast<uint4*>(o + 8ull * i);
    po[0] = make_uint4(w[0], w[1], w[2], w[3]);
    po[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// o[i] = a[i] b[i]  (hadamard, util/mod.rs:80-85); o may alias a or b
template <int SF>
__global__ void __launch_bounds__(256) k_hadamard(const uint32_t* a, const uint32_t* b, uint32_t* o, uint32_t n) {
  using F = Fp<SF>;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint4* pa = reinterpret_cast<const uint4*>(a + 8ull * i);
    const uint4* pb = reinterpret_cast<const uint4*>(b + 8ull * i);
    uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    uint32_t wa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    uint32_t wb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    // montmul(a 2^256, b 2^261) = ab 2^256: ark form again
    uint32_t w[8];
    F::store_packed(F::mul(F::unpack(wa), F::load_ark(wb)), w);
    uint4* po = reinterpret_cast<uint4*>(o + 8ull * i);
    po[0] = make_uint4(w[0], w[1], w[2], w[3]);
    po[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

struct PowersArgs {
  uint32_t f[8];  // ark form
  uint64_t start;
};
// o[i] = f^(start + i)  (powers, util/mod.rs:57-64; f^0 = 1 also for f = 0, as Field::pow).  A lane
// raises f to start + (its first index) by square-and-multiply and then walks its CHUNK consecutive
// outputs with one multiplication each.
template <int SF>
__global__ void __launch_bounds__(256) k_powers(PowersArgs args, uint32_t* __restrict__ o, uint32_t n) {
  using F = Fp<SF>;
  constexpr uint32_t CHUNK = 8;
  const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * CHUNK;
  if (first >= n) return;
  const Fe f = F::load_ark(args.f);
  const uint64_t e = args.start + first;
  Fe r = F::one();
  for (int b = 63; b >= 0; b--) {
    r = F::sqr(r);
    if ((e >> b) & 1ull) r = F::mul(r, f);
  }
  for (uint32_t t = 0; t < CHUNK && first + t < n; t++) {
    uint32_t w[8];
    F::store_ark(r, w);
    uint4* po = reinterpret_cast<uint4*>(o + 8ull * (first + t));
    po[0] = make_uint4(w[0], w[1], w[2], w[3]);
    po[1] = make_uint4(w[4], w[5], w[6], w[7]);
    r = F::mul(r, f);
  }
}

// o[i] = s[i] g[i]  (hadamard_gp, util/mod.rs:70-78, used at com_r1cs_to_ipa.rs:110: every lane has its own multiplier).
// Every lane runs the same schedule; only table indices and signs differ:
//   * GLV on the device: s = k1 + k2 lambda with |k1|, |k2| < 2^129 (glv_split), so the chain is 132 doublings long
//     instead of 255, with phi(P) = (zeta x, y) as the second base;
//   * signed 3-bit windows without a carry chain: k' = |k| + sum_j 4 * 8^j, digit_j = (bits 3j..3j+2 of k') - 4 in
//     [-4, 3]; 44 windows cover k' < 2^132;
//   * the multiples 2P, 3P, 4P are made once per lane (two doublings, one addition, in Jacobian), brought to affine with
//     ONE inversion shared by the workgroup, and kept in LDS as unpacked limbs; P itself stays in registers;
//   * the chain runs in Jacobian coordinates (3 doublings + at most two mixed additions per window); the inversion
//     back to affine is again the workgroup's.
// 132 doublings + about 77 additions per output, against 255 + 127 for the plain ladder.  o may alias g.
constexpr int HAD_WINDOWS = 44;
template <int BF>
__global__ void __launch_bounds__(256, 2) k_hadamard_points(const uint32_t* g, const uint32_t* __restrict__ s, uint32_t* o,
                                                            uint32_t n) {
  using F = Fp<BF>;
  using FS = Fp<1 - BF>;
  using E = Ec<BF>;
  using IO = PointIo<BF>;
  __shared__ uint32_t tree[512][9];
  __shared__ int32_t tab[3][18][256];  // 2P, 3P, 4P of this lane: x (9 limbs), y (9 limbs)
  const uint32_t tid = threadIdx.x;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  Aff P;
  P.inf = true;
  P.x = F::zero();
  P.y = F::zero();
  uint64_t k1[4] = {0, 0, 0, 0}, k2[4] = {0, 0, 0, 0};
  bool n1 = false, n2 = false;
  if (active) {
    P = IO::load_dev_affine(g + 16ull * i);
    uint32_t w[8];
#pragma unroll
    for (int t = 0; t < 8; t++) w[t] = s[8ull * i + t];
    FS::pack(FS::canon(FS::mul(FS::unpack(w), FS::from_const(FS::C::C256TOCANON))), w);
    const uint64_t k[4] = {w[0] | ((uint64_t)w[1] << 32), w[2] | ((uint64_t)w[3] << 32), w[4] | ((uint64_t)w[5] << 32),
                           w[6] | ((uint64_t)w[7] << 32)};
    glv_split<BF>(k, k1, n1, k2, n2);
    // k' = |k| + 0x4924...: every window offset by 4 (0b100 repeated: 0x9249 2492 4924 ... in 64-bit words)
    const uint64_t off[3] = {0x4924924924924924ull, 0x2492492492492492ull, 0x9ull};  // bits 3j+2 set for j < 44
    unsigned __int128 c = 0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      c += (unsigned __int128)k1[t] + off[t];
      k1[t] = (uint64_t)c;
      c >>= 64;
    }
    c = 0;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      c += (unsigned __int128)k2[t] + off[t];
      k2[t] = (uint64_t)c;
      c >>= 64;
    }
  }
  const bool fin0 = active && !P.inf;
  // the table: 2P, 3P, 4P in affine through one shared inversion
  {
    Jac P2 = E::jidentity(), P3 = E::jidentity(), P4 = E::jidentity();
    if (fin0) {
      P2 = E::jdbl(E::jfrom_affine(P));
      P3 = P2;
      E::jmadd(P3, P);
      P4 = E::jdbl(P2);
    }
    // on a prime-order curve none of 2P, 3P, 4P is the identity for a finite P
    const Fe z2 = fin0 ? P2.z : F::one(), z3 = fin0 ? P3.z : F::one(), z4 = fin0 ? P4.z : F::one();
    const Fe z23 = F::mul(z2, z3);
    const Fe iv = block_inverse_256<BF>(F::mul(z23, z4), tree);
    const Fe i4 = F::mul(iv, z23), i23 = F::mul(iv, z4);
    const Fe i2 = F::mul(i23, z3), i3 = F::mul(i23, z2);
#pragma unroll 1
    for (int t = 0; t < 3; t++) {
      const Jac& J = t == 0 ? P2 : (t == 1 ? P3 : P4);
      const Fe& zi = t == 0 ? i2 : (t == 1 ? i3 : i4);
      const Aff A = E::jto_affine_with(J, zi);
      const Fe ay = F::carry(A.y);
#pragma unroll
      for (int k = 0; k < 9; k++) {
        tab[t][k][tid] = A.x.l[k];
        tab[t][9 + k][tid] = ay.l[k];
      }
    }
  }
  Jac acc = E::jidentity();
  if (fin0) {
    const Fe zeta = F::from_const(CurveConsts<BF>::ZETA261);
#pragma unroll 1
    for (int w = HAD_WINDOWS - 1; w >= 0; w--) {
      acc = E::jdbl(E::jdbl(E::jdbl(acc)));
      const int pos = 3 * w, wi = pos >> 6, sh = pos & 63;
#pragma unroll 1
      for (int h = 0; h < 2; h++) {
        const uint64_t lo = h == 0 ? (wi == 0 ? k1[0] : (wi == 1 ? k1[1] : k1[2])) : (wi == 0 ? k2[0] : (wi == 1 ? k2[1] : k2[2]));
        const uint64_t hi = h == 0 ? (wi == 0 ? k1[1] : (wi == 1 ? k1[2] : 0ull)) : (wi == 0 ? k2[1] : (wi == 1 ? k2[2] : 0ull));
        uint32_t bits = (uint32_t)(lo >> sh);
        if (sh > 61) bits |= (uint32_t)(hi << (64 - sh));
        const int d = (int)(bits & 7u) - 4;
        const int mag = d < 0 ? -d : d;
        Aff t;
        t.inf = mag == 0;
        t.x = P.x;
        t.y = P.y;
        if (mag >= 2) {
#pragma unroll
          for (int k = 0; k < 9; k++) {
            t.x.l[k] = tab[mag - 2][k][tid];
            t.y.l[k] = tab[mag - 2][9 + k][tid];
          }
        }
        if (h == 1) t.x = F::mul(t.x, zeta);  // phi(T) = (zeta x, y)
        const bool neg = (d < 0) != (h == 0 ? n1 : n2);
        if (neg) t.y = F::neg(t.y);
        E::jmadd(acc, t);  // skips a zero digit (t.inf)
      }
    }
  }
  const bool fin = active && !acc.inf;
  const Fe inv = block_inverse_256<BF>(fin ? acc.z : F::one(), tree);
  if (active) IO::store_dev_affine(o + 16ull * i, E::jto_affine_with(acc, inv));
}

// Verifier scalars of Bp::verify (bp_iter.rs:25-42,160-164), 2 np of them:
//   o[i]      = final_a * s_i,      s_i     = prod_t (bit t of i ? x_{K-1-t} : x_{K-1-t}^-1)
//   o[np + i] = final_b * s_inv_i,  s_inv_i = the same product over the inverted challenges
// written as base * prod_{set bits} (square of the challenge):  tab = [base_a, base_b, sq_0 .. sq_{K-1},
// sqi_0 .. sqi_{K-1}] (ark form, device memory) with base_a = final_a prod x^-1, sq_t = x_{K-1-t}^2,
// base_b = final_b prod x, sqi_t = x_{K-1-t}^-2.  blockIdx.y: 0 = the a half, 1 = the b half.
template <int SF>
__global__ void __launch_bounds__(256) k_verify_scalars(const uint32_t* __restrict__ tab, uint32_t* __restrict__ o,
                                                        uint32_t np, int K) {
  using F = Fp<SF>;
  __shared__ int32_t sq[32 * 9];
  const uint32_t half = blockIdx.y;
  if (threadIdx.x < (uint32_t)K) {
    uint32_t w[8];
#pragma unroll
    for (int t = 0; t < 8; t++) w[t] = tab[8ull * (2 + half * K + threadIdx.x) + t];
    const Fe m = F::load_ark(w);
#pragma unroll
    for (int t = 0; t < 9; t++) sq[9 * threadIdx.x + t] = m.l[t];
  }
  __syncthreads();
  uint32_t wb[8];
#pragma unroll
  for (int t = 0; t < 8; t++) wb[t] = tab[8ull * half + t];
  const Fe base = F::unpack(wb);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
    Fe s = base;
    for (int t = 0; t < K; t++) {
      if ((i >> t) & 1u) {
        Fe m;
#pragma unroll
        for (int u = 0; u < 9; u++) m.l[u] = sq[9 * t + u];
        s = F::mul(s, m);  // ark form times device form stays ark form
      }
    }
    uint32_t w[8];
    F::store_packed(s, w);
    uint4* po = reinterpret_cast<uint4*>(o + 8ull * ((uint64_t)half * np + i));
    po[0] = make_uint4(w[0], w[1], w[2], w[3]);
    po[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// ---------------------------------------------------------------------------
// Sparse matrix - vector products of the Com-R1CS compiler: mat_vec_mult and
// vec_mat_mult, src/relations/com_r1cs.rs:43-57, :67-83, as called at
// src/reductions/com_r1cs_to_ipa.rs:63-64, 93-99.  One kernel serves both: the matrix arrives in
// compressed rows,
//     out[r] = sum_{e in [ptr[r], ptr[r+1])} val[perm ? perm[e] : e] * vec[idx[e]],
// and the transposed product runs it on the column-compressed form the host derives (perm then points
// back into the caller's value array, so the 32-byte values are never reordered).  SPMV_GROUP lanes
// share a row: R1CS rows hold a handful of entries, with a few long ones.  Scalars in ark form.
// ---------------------------------------------------------------------------
// GROUP = 8 for matrices with long rows; 2 when every row is short (the R1CS shape: a handful of entries a row -- with
// eight lanes a row most lanes idle through the multiplication and the kernel is bound by its instruction issue).
template <int SF, uint32_t SPMV_GROUP>
__global__ void __launch_bounds__(256) k_spmv(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx,
                                              const uint32_t* __restrict__ perm, const uint32_t* __restrict__ val,
                                              const uint32_t* __restrict__ vec, uint32_t* __restrict__ out, uint32_t n_rows) {
  using F = Fp<SF>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = t / SPMV_GROUP, sub = t % SPMV_GROUP;
  Fe acc = F::zero();
  if (r < n_rows) {  // no early return: the shuffles below need every lane of the wave
    const uint32_t lo = ptr[r], hi = ptr[r + 1];
    for (uint32_t e = lo + sub; e < hi; e += SPMV_GROUP) {
      const uint32_t ve = perm ? perm[e] : e;
      const uint4* pa = reinterpret_cast<const uint4*>(val + 8ull * ve);
      const uint4* pb = reinterpret_cast<const uint4*>(vec + 8ull * idx[e]);
      uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
      uint32_t wa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      uint32_t wb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      // (a 2^256)(b 2^256)/2^261 = ab 2^251, fixed up once per row below
      acc = fe_partial_reduce<SF>(F::add(acc, F::mul(F::unpack(wa), F::unpack(wb))));
    }
  }
#pragma unroll
  for (uint32_t off = SPMV_GROUP / 2; off >= 1; off >>= 1) {
    Fe o;
#pragma unroll
    for (int k = 0; k < 9; k++) o.l[k] = __shfl_down(acc.l[k], off, 64);
    acc = fe_partial_reduce<SF>(F::add(acc, o));
  }
  if (r < n_rows && sub == 0) {
    uint32_t w[8];
    F::store_packed(F::mul(acc, F::from_const(F::C::C256TO261)), w);  // x 2^251 -> x 2^256
    uint4* po = reinterpret_cast<uint4*>(out + 8ull * r);
    po[0] = make_uint4(w[0], w[1], w[2], w[3]);
    po[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// ---------------------------------------------------------------------------
// Seed-faithful point sampling: the data-parallel half of ark-ec 0.3's
// GroupProjective::rand for y^2 = x^3 + 5, which examples/msm_bench.rs:41 and
// src/relations/ipa.rs:28-36 draw their points with.  The host walks the RNG stream (the draws of an
// attempt -- x by rejection, then one bool -- do not depend on the square root's outcome) and hands
// every candidate here: ok[i] = 1 iff x^3 + 5 is a square, and then y[i] is the root ark picks,
// (y < -y) XOR greatest ? y : -y on canonical integers.  Tonelli-Shanks for p - 1 = 2^32 t; `root` is a
// 2^32-th root of unity of full order (any: the result does not depend on it).  BF = base field.
// ---------------------------------------------------------------------------
struct LiftArgs {
  uint32_t root[8];  // ark form
};
template <int BF>
__global__ void __launch_bounds__(256) k_lift_x(LiftArgs args, const uint32_t* __restrict__ xs,
                                                const uint8_t* __restrict__ greatest, uint32_t* __restrict__ ys,
                                                uint8_t* __restrict__ ok, uint32_t n) {
  using F = Fp<BF>;
  using C = typename F::C;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  {
    const uint4* px = reinterpret_cast<const uint4*>(xs + 8ull * i);
    uint4 a0 = px[0], a1 = px[1];
    w[0] = a0.x; w[1] = a0.y; w[2] = a0.z; w[3] = a0.w; w[4] = a1.x; w[5] = a1.y; w[6] = a1.z; w[7] = a1.w;
  }
  const Fe one = F::one();
  const Fe x = F::load_ark(w);
  Fe five = F::carry(F::add(one, one));
  five = F::carry(F::add(F::carry(F::add(five, five)), one));
  const Fe a = F::carry(F::add(F::mul(F::sqr(x), x), five));
  auto is_one = [&](const Fe& v) { return F::is_zero(F::sub(v, one)); };
  bool good = true;
  Fe y = F::zero();
  if (!F::is_zero(a)) {
    // a^((t-1)/2), t = (p - 1) >> 32 = words 1..7 of p (odd), so (t-1)/2 = t >> 1: 223 bits
    Fe wv = one;
    for (int b = 222; b >= 0; b--) {
      const int bit = b + 1;  // bit of t
      const uint32_t word = C::P32[1 + (bit >> 5)];
      wv = F::sqr(wv);
      if ((word >> (bit & 31)) & 1u) wv = F::mul(wv, a);
    }
    y = F::mul(a, wv);        // a^((t+1)/2)
    Fe b = F::mul(y, wv);     // a^t
    Fe z = F::load_ark(args.root);
    int v = 32;
    for (int guard = 0; guard < 33 && good && !is_one(b); guard++) {
      int k = 0;
      Fe b2k = b;
      while (k < v && !is_one(b2k)) {
        b2k = F::sqr(b2k);
        k++;
      }
      if (k >= v) {
        good = false;  // b has order 2^32 or more in a group of order 2^v: not a square
        break;
      }
      Fe ww = z;
      for (int j = 0; j < v - k - 1; j++) ww = F::sqr(ww);
      z = F::sqr(ww);
      b = F::mul(b, z);
      y = F::mul(y, ww);
      v = k;
    }
    if (good && !is_one(b)) good = false;
  }
  uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (good) {
    // canonical integer of y: montmul(y 2^261, 1) = y; y < p - y  <=>  y <= (p - 1) / 2
    Fe plain = F::zero();
    plain.l[0] = 1;
    uint32_t yi[8];
    F::store_packed(F::mul(y, plain), yi);
    bool y_less = true;  // equality cannot happen for y != 0 (p odd); y = 0: both roots are 0
    for (int k = 7; k >= 0; k--) {
      const uint32_t h = (C::P32[k] >> 1) | (k < 7 ? C::P32[k + 1] << 31 : 0u);
      if (yi[k] != h) {
        y_less = yi[k] < h;
        break;
      }
    }
    const bool g = greatest[i] != 0;
    F::store_ark((y_less != g) ? y : F::neg(y), o);
  }
  uint4* py = reinterpret_cast<uint4*>(ys + 8ull * i);
  py[0] = make_uint4(o[0], o[1], o[2], o[3]);
  py[1] = make_uint4(o[4], o[5], o[6], o[7]);
  ok[i] = good ? 1 : 0;
}

// Register-only multiply-add loop: sustained v_mad_u64_u32 issue rate (the integer
// roofline the MSM and the generator fold are priced against).
__global__ void __launch_bounds__(256) k_mad_peak(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a = seed + threadIdx.x, b = seed * 3u + blockIdx.x;
  uint64_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = a + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i];
  if (s == 0x1234567) out[0] = (uint32_t)s;
}

}  // namespace bpgpu
