// Dependent-issue microbenchmark for gfx950: how many independent v_mad_i64_i32 chains does a wave need
// to keep its SIMD's issue slots full at 1, 2, 4 waves per SIMD?  (Decides whether a field multiplication
// may run as ONE accumulator chain -- product scanning -- or must interleave columns.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 16384;
constexpr int PER_ITER = 48;

// CHAINS independent accumulators, PER_ITER mads per iteration dealt round-robin
template <int CHAINS>
__global__ void __launch_bounds__(256) k_chain(uint32_t* out, uint32_t seed) {
  int32_t a = (int32_t)(seed + threadIdx.x), b = (int32_t)(seed * 3 + blockIdx.x);
  int64_t acc[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) acc[i] = a + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int u = 0; u < PER_ITER; u++)
      asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[u % CHAINS]) : "v"(a), "v"(b) : "vcc");
  }
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += acc[i];
  if (s == 0x1234567) out[0] = (uint32_t)s;
}

// the column pattern of a product-scanning Montgomery multiplication: 12 dependent mads, then and + 64-bit
// arithmetic shift on the same accumulator; CHAINS such columns interleaved
template <int CHAINS>
__global__ void __launch_bounds__(256) k_column(uint32_t* out, uint32_t seed) {
  int32_t a = (int32_t)(seed + threadIdx.x), b = (int32_t)(seed * 3 + blockIdx.x);
  int64_t acc[CHAINS];
  uint32_t m[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) { acc[i] = a + i; m[i] = 0; }
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int col = 0; col < 4; col++) {
#pragma unroll
      for (int u = 0; u < 12 * CHAINS; u++)
        asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[u % CHAINS]) : "v"(a), "v"(b) : "vcc");
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        asm volatile("v_and_b32 %0, 0x1fffffff, %1" : "=v"(m[c]) : "v"((uint32_t)acc[c]));
        asm volatile("v_ashrrev_i64 %0, 29, %0" : "+v"(acc[c]));
        a ^= (int32_t)m[c];
      }
    }
  }
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += acc[i] + m[i];
  if (s == 0x1234567) out[0] = (uint32_t)s;
}

int g_block = 256;
template <class K>
void run(const char* name, K kern, int blocks_per_cu, double instr_per_lane, uint32_t* dout) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int grid = prop.multiProcessorCount * blocks_per_cu;
  const int block = g_block;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  kern<<<grid, block>>>(dout, 1); CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < 4; r++) {
    CHECK(hipEventRecord(e0));
    kern<<<grid, block>>>(dout, r + 2);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double ops = (double)grid * block * instr_per_lane;
  printf("%-34s block=%d blocks/CU=%d  %8.3f ms  %7.3f T lane-ops/s\n", name, block, blocks_per_cu, best, ops / (best * 1e-3) / 1e12);
}

int main() {
  uint32_t* dout; CHECK(hipMalloc(&dout, 4096));
  for (int bpc : {1, 2, 3, 4}) {
    run("mad chain x1", k_chain<1>, bpc, (double)ITERS * PER_ITER, dout);
    run("mad chain x2", k_chain<2>, bpc, (double)ITERS * PER_ITER, dout);
    run("mad chain x3", k_chain<3>, bpc, (double)ITERS * PER_ITER, dout);
    run("mad chain x4", k_chain<4>, bpc, (double)ITERS * PER_ITER, dout);
    run("mad chain x8", k_chain<8>, bpc, (double)ITERS * PER_ITER, dout);
    run("column (12 mad + and + ashr) x1", k_column<1>, bpc, (double)ITERS * 4 * 14, dout);
    run("column (12 mad + and + ashr) x2", k_column<2>, bpc, (double)ITERS * 4 * 14 * 2, dout);
  }
  // one-wave workgroups: where do the waves of a CU land, and how fast does a lone wave issue?
  g_block = 64;
  for (int bpc : {1, 2, 4, 8, 16}) {
    run("mad chain x1", k_chain<1>, bpc, (double)ITERS * PER_ITER, dout);
    run("mad chain x4", k_chain<4>, bpc, (double)ITERS * PER_ITER, dout);
  }
  return 0;
}
