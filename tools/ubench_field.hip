// Field-multiplication throughput microbenchmark (gfx950): chains of Fp::mul / Fp::sqr in
// registers at several occupancies, to price the real instruction stream against the VALU
// issue peak.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bulletproofs_amd/csrc tools/ubench_field.hip -o tools/ubench_field
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "curve.cuh"
using namespace bpgpu;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(64) k_mul(uint32_t* out, int iters, uint32_t seed) {
  using F = Fp<0>;
  Fe x[CHAINS], y;
  for (int c = 0; c < CHAINS; c++)
    for (int i = 0; i < 9; i++) x[c].l[i] = (int32_t)((seed * 2654435761u + threadIdx.x * 40503u + i * 977u + c * 131u) & MASK29);
  for (int i = 0; i < 9; i++) y.l[i] = (int32_t)((seed * 97u + blockIdx.x * 31u + i * 7919u) & MASK29);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (MODE == 0) x[c] = F::mul(x[c], y);
      if (MODE == 1) x[c] = F::sqr(x[c]);
    }
  }
  uint32_t s = 0;
  for (int c = 0; c < CHAINS; c++)
    for (int i = 0; i < 9; i++) s += (uint32_t)x[c].l[i];
  if (s == 0x12345) out[0] = s;
}

template <int MODE, int CHAINS>
void run(const char* name, int waves_per_simd, uint32_t* dout) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int grid = prop.multiProcessorCount * 4 * waves_per_simd;
  int iters = 4096;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_mul<MODE, CHAINS><<<grid, 64>>>(dout, iters, 1); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; r++) {
    CHECK(hipEventRecord(e0));
    k_mul<MODE, CHAINS><<<grid, 64>>>(dout, iters, r + 2);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double muls = (double)grid * 64 * iters * CHAINS;
  double ns_per_wave_mul_per_simd = best * 1e6 / ((double)iters * CHAINS * waves_per_simd);
  printf("%-10s chains=%d waves/SIMD=%d  %8.3f ms  %7.2f Gmul/s  %7.1f ns per wave-mul per SIMD (= %5.0f cycles at 2.27 GHz)\n", name, CHAINS,
         waves_per_simd, best, muls / best / 1e6, ns_per_wave_mul_per_simd, ns_per_wave_mul_per_simd * 2.27);
}

int main() {
  uint32_t* dout; CHECK(hipMalloc(&dout, 4096));
  for (int w : {1, 2, 3, 4, 8}) {
    run<0, 1>("mul", w, dout);
    run<0, 2>("mul", w, dout);
    run<1, 1>("sqr", w, dout);
  }
  return 0;
}
