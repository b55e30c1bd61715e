// Batched-affine bucket accumulation, priced on the hardware ("Next" 1a): the MOST favourable on-chip form --
// every lane owns B buckets whose affine accumulators live in registers, every step adds one gathered point to each of
// them, the B denominators of a lane are chained (Montgomery's trick) and the workgroup's 256 lane products share ONE
// inversion (block_inverse_256, the product tree + single-lane safegcd root the fold kernels use) -- against the XYZZ
// mixed addition k_accum runs (Ec::madd), on the same gathered points, same grid shapes.
//
//   per addition, batched affine:  dx, 1 M prefix, 2 M to peel the inverse, 1 M slope, 1 S, 1 M  = 5 M + 1 S
//                                  + (tree: 3 M per lane + 16 barriers + a 13 K-slot root inversion on ONE lane) / B
//   per addition, XYZZ mixed:      8 M + 2 S (Y3's two products share a reduction)
//
// Special cases (equal x: doubling / cancellation, identity operands) are left out of the affine kernel: a real one needs
// them, which only adds to its cost.  Correctness of the affine step is checked against Ec::madd + to_affine first.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bulletproofs_amd/csrc tools/ubench_batch_affine.hip -o /tmp/ubench_batch_affine
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.cuh"
using namespace bpgpu;
#define CHECK(x)                                                                       \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

constexpr int BF = 0;
using F = Fp<BF>;
using E = Ec<BF>;
using IO = PointIo<BF>;

// pts[i] = (i + 2) G for the generator G = (-1, 2) of Pallas: distinct finite points
__global__ void __launch_bounds__(256) k_make_points(uint32_t* pts, uint32_t n) {
  __shared__ uint32_t tree[512][9];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  Aff G;
  G.inf = false;
  G.x = F::carry(F::neg(F::one()));
  G.y = F::carry(F::add(F::one(), F::one()));
  uint32_t k[8] = {i + 2, 0, 0, 0, 0, 0, 0, 0};
  Xyzz acc = E::identity();
  for (int b = 31; b >= 0; b--) {
    acc = E::dbl(acc);
    if ((k[0] >> b) & 1u) E::madd(acc, G);
  }
  const Fe inv = block_inverse_256<BF>(acc.inf ? F::one() : F::mul(acc.zz, acc.zzz), tree);
  Aff r;
  r.inf = acc.inf;
  r.x = F::mul(acc.x, F::mul(inv, acc.zzz));
  r.y = F::mul(acc.y, F::mul(inv, acc.zz));
  if (i < n) IO::store_dev_affine(pts + 16ull * i, r);
}

// B accumulators per lane, `steps` additions into each; gathers are pseudo-random 64-byte reads like k_accum's
template <int B>
__global__ void __launch_bounds__(256, (B <= 4 ? 2 : 1)) k_affine(const uint32_t* __restrict__ pts, uint32_t npts, uint32_t* __restrict__ out,
                                                                  int steps) {
  __shared__ uint32_t tree[512][9];
  const uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
  Fe ax[B], ay[B];
#pragma unroll
  for (int j = 0; j < B; j++) {
    const Aff p = IO::load_dev_affine(pts + 16ull * ((lane_id * B + j) % npts));
    ax[j] = p.x;
    ay[j] = p.y;
  }
  uint32_t idx = lane_id * 2654435761u;
  for (int s = 0; s < steps; s++) {
    Fe qx[B], qy[B], pre[B];
    Fe run = F::one();
#pragma unroll
    for (int j = 0; j < B; j++) {
      idx = idx * 1664525u + 1013904223u;
      const Aff q = IO::load_dev_affine(pts + 16ull * ((idx >> 8) % npts));
      qx[j] = q.x;
      qy[j] = q.y;
      pre[j] = run;
      run = F::mul(run, F::sub(q.x, ax[j]));  // prefix products of the denominators
    }
    Fe inv = block_inverse_256<BF>(run, tree);
#pragma unroll
    for (int j = B - 1; j >= 0; j--) {
      const Fe dx = F::sub(qx[j], ax[j]);
      const Fe di = F::mul(inv, pre[j]);  // 1 / dx_j
      inv = F::mul(inv, dx);              // drop dx_j from the running inverse
      const Fe lam = F::mul(F::sub(qy[j], ay[j]), di);
      const Fe x3 = F::carry(F::sub(F::sub(F::sqr(lam), ax[j]), qx[j]));
      ay[j] = F::carry(F::sub(F::mul(lam, F::sub(ax[j], x3)), ay[j]));
      ax[j] = x3;
    }
  }
#pragma unroll
  for (int j = 0; j < B; j++) {
    Aff r;
    r.inf = false;
    r.x = ax[j];
    r.y = ay[j];
    IO::store_dev_affine(out + 16ull * ((uint64_t)lane_id * B + j), r);
  }
}

// the same schedule with XYZZ accumulators and Ec::madd (what k_accum does per entry); normalised at the end for the check
template <int B>
__global__ void __launch_bounds__(256, (B <= 1 ? 2 : 1)) k_xyzz(const uint32_t* __restrict__ pts, uint32_t npts, uint32_t* __restrict__ out, int steps,
                                                                int normalise) {
  __shared__ uint32_t tree[512][9];
  const uint32_t lane_id = blockIdx.x * blockDim.x + threadIdx.x;
  Xyzz acc[B];
#pragma unroll
  for (int j = 0; j < B; j++) acc[j] = E::from_affine(IO::load_dev_affine(pts + 16ull * ((lane_id * B + j) % npts)));
  uint32_t idx = lane_id * 2654435761u;
  for (int s = 0; s < steps; s++) {
#pragma unroll
    for (int j = 0; j < B; j++) {
      idx = idx * 1664525u + 1013904223u;
      E::madd(acc[j], IO::load_dev_affine(pts + 16ull * ((idx >> 8) % npts)));
    }
  }
#pragma unroll 1
  for (int j = 0; j < B; j++) {
    if (normalise) {
      const Fe inv = block_inverse_256<BF>(acc[j].inf ? F::one() : F::mul(acc[j].zz, acc[j].zzz), tree);
      Aff r;
      r.inf = acc[j].inf;
      r.x = F::mul(acc[j].x, F::mul(inv, acc[j].zzz));
      r.y = F::mul(acc[j].y, F::mul(inv, acc[j].zz));
      IO::store_dev_affine(out + 16ull * ((uint64_t)lane_id * B + j), r);
    } else {
      IO::store_dev_xyzz(out + 32ull * ((uint64_t)lane_id * B + j), acc[j]);
    }
  }
}

// one F::inv per lane (all 64 lanes of a wave run the same schedule): latency of the root inversion
__global__ void __launch_bounds__(64) k_inv_latency(const uint32_t* pts, uint32_t* out, int reps) {
  Fe v = IO::load_dev_affine(pts + 16ull * threadIdx.x).x;
  for (int r = 0; r < reps; r++) v = F::carry(F::add(F::inv(v), F::one()));
  uint32_t w[8];
  F::store_packed(v, w);
  for (int k = 0; k < 8; k++) out[8 * threadIdx.x + k] = w[k];
}

template <class L>
float time_ms(L launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

template <int B>
void bench_affine(const uint32_t* pts, uint32_t npts, uint32_t* out, int cus, int wgs_per_cu) {
  const int steps = 64, grid = cus * wgs_per_cu;
  const float ms = time_ms([&] { k_affine<B><<<grid, 256>>>(pts, npts, out, steps); });
  const double adds = (double)grid * 256 * B * steps;
  printf("batched affine  B=%d  %d WG/CU (%d waves/SIMD)  %8.3f ms  %7.2f G additions/s   %d additions per inversion\n", B, wgs_per_cu,
         wgs_per_cu, ms, adds / ms / 1e6, 256 * B);
}
template <int B>
void bench_xyzz(const uint32_t* pts, uint32_t npts, uint32_t* out, int cus, int wgs_per_cu) {
  const int steps = 64, grid = cus * wgs_per_cu;
  const float ms = time_ms([&] { k_xyzz<B><<<grid, 256>>>(pts, npts, out, steps, 0); });
  const double adds = (double)grid * 256 * B * steps;
  printf("XYZZ mixed add  B=%d  %d WG/CU (%d waves/SIMD)  %8.3f ms  %7.2f G additions/s\n", B, wgs_per_cu, wgs_per_cu, ms,
         adds / ms / 1e6);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const uint32_t npts = 1u << 20;
  uint32_t *pts, *out, *out2;
  CHECK(hipMalloc(&pts, (size_t)npts * 64));
  CHECK(hipMalloc(&out, (size_t)cus * 2 * 256 * 8 * 128));
  CHECK(hipMalloc(&out2, (size_t)cus * 2 * 256 * 8 * 128));
  k_make_points<<<npts / 256, 256>>>(pts, npts);
  CHECK(hipDeviceSynchronize());
  // correctness of the batched affine step: 8 steps, B = 4, against madd + normalisation
  {
    k_affine<4><<<4, 256>>>(pts, npts, out, 8);
    k_xyzz<4><<<4, 256>>>(pts, npts, out2, 8, 1);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> a(4 * 256 * 4 * 16), b(a.size());
    CHECK(hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), out2, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
    printf("affine step vs madd + normalise on %zu points: %s\n", a.size() / 16, bad ? "MISMATCH" : "identical");
    if (bad) return 1;
  }
  {
    const int reps = 16;
    const float ms = time_ms([&] { k_inv_latency<<<1, 64>>>(pts, out, reps); });
    printf("field inversion (safegcd, lone wave): %.2f us each\n", ms * 1e3 / reps);
  }
  bench_xyzz<1>(pts, npts, out, cus, 2);
  bench_xyzz<1>(pts, npts, out, cus, 1);
  bench_affine<1>(pts, npts, out, cus, 2);
  bench_affine<2>(pts, npts, out, cus, 2);
  bench_affine<4>(pts, npts, out, cus, 2);
  bench_affine<4>(pts, npts, out, cus, 1);
  bench_affine<8>(pts, npts, out, cus, 1);
  return 0;
}
