// Mixed-add throughput microbenchmark (gfx950): Ec::madd from registers (no memory), to separate
// instruction-issue limits from gather latency in k_accum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "curve.cuh"
using namespace bpgpu;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k_madd(uint32_t* out, const uint32_t* pts, int iters, uint32_t seed) {
  using E = Ec<0>;
  using IO = PointIo<0>;
  Aff q0 = IO::load_dev_affine(pts + 16 * (threadIdx.x & 7));
  Aff q1 = IO::load_dev_affine(pts + 16 * ((threadIdx.x + seed) & 7) + 128);
  Xyzz acc = E::identity();
  if (MODE == 3) {  // desynchronise the waves: a per-wave pseudo-random number of squarings first
    int spin = (int)((blockIdx.x * 2654435761u) >> 25);  // 0..127
    Fe t = q0.x;
    for (int i = 0; i < spin; i++) t = Fp<0>::sqr(t);
    q0.x = t;
  }
  for (int it = 0; it < iters; it++) {
    if (MODE == 0 || MODE == 3) {
      E::madd(acc, (it & 1) ? q0 : q1);
    } else if (MODE == 1) {
      acc = E::dbl(acc);
      E::madd(acc, q0);
    } else {
      Aff q = IO::load_dev_affine(pts + 16 * ((threadIdx.x * 7 + it * 13 + blockIdx.x) & 15));
      E::madd(acc, q);
    }
  }
  IO::store_dev_xyzz(out + 32ull * (blockIdx.x * 64 + threadIdx.x), acc);
}

template <int MODE>
void run(const char* name, int waves_per_simd, uint32_t* dout, uint32_t* dpts) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int grid = prop.multiProcessorCount * 4 * waves_per_simd;
  int iters = 256;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_madd<MODE><<<grid, 64>>>(dout, dpts, iters, 1); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0));
    k_madd<MODE><<<grid, 64>>>(dout, dpts, iters, r + 2);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double ns = best * 1e6 / ((double)iters * waves_per_simd);
  printf("%-12s waves/SIMD=%d  %8.3f ms  %7.1f ns per wave-iter per SIMD (= %6.0f cycles at 2.27 GHz), %7.2f G iter/s\n", name, waves_per_simd,
         best, ns, ns * 2.27, (double)grid * 64 * iters / best / 1e6);
}

int main() {
  uint32_t *dout, *dpts;
  CHECK(hipMalloc(&dout, 256 * 4 * 8 * 64 * 128));
  CHECK(hipMalloc(&dpts, 4096));
  // arbitrary "points" (not on the curve: the formulas do not care)
  uint32_t h[1024];
  for (int i = 0; i < 1024; i++) h[i] = (i * 2654435761u) >> 3;
  CHECK(hipMemcpy(dpts, h, 4096, hipMemcpyHostToDevice));
  for (int w : {1, 2, 3, 4}) {
    run<0>("madd(reg)", w, dout, dpts);
    run<2>("madd(L1 ld)", w, dout, dpts);
    run<3>("madd desync", w, dout, dpts);
  }
  return 0;
}
