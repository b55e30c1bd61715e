// Issue rates of the carry-handling instructions of the 29-bit-limb Montgomery multiplication (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 16384;
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
  uint64_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = ((uint64_t)a << 32) + b + i;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        uint32_t lo = (uint32_t)acc[i];
        if (MODE == 0) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(acc[i]));
        if (MODE == 1) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i]));
        if (MODE == 2) { asm volatile("v_alignbit_b32 %0, %1, %0, 29" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        if (MODE == 3) { asm volatile("v_bfe_i32 %0, %0, 0, 29" : "+v"(lo)); acc[i] = lo; }
        if (MODE == 4) { asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(lo)); acc[i] = lo; }
        if (MODE == 5) asm volatile("v_mad_i64_i32 %0, vcc, %1, 8, %0" : "+v"(acc[i]) : "v"(a) : "vcc");
        if (MODE == 6) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
        if (MODE == 7) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc[i]));
        if (MODE == 8) { asm volatile("v_ashrrev_i32 %0, 29, %0" : "+v"(lo)); acc[i] = lo; }
        if (MODE == 9) { asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        if (MODE == 10) { asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(lo) : "v"(a)); acc[i] = lo; }
        if (MODE == 11) { asm volatile("v_add3_u32 %0, %1, %0, %2" : "+v"(lo) : "v"(a), "v"(b)); acc[i] = lo; }
        if (MODE == 12) { asm volatile("v_lshl_add_u32 %0, %1, 3, %0" : "+v"(lo) : "v"(a)); acc[i] = lo; }
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i];
  if (s == 0x1234567) out[0] = (uint32_t)s;
}
template <int MODE>
void run(const char* name, int bpc, uint32_t* dout) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int grid = prop.multiProcessorCount * bpc;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<MODE><<<grid, 256>>>(dout, 1); CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < 4; r++) {
    CHECK(hipEventRecord(e0)); k<MODE><<<grid, 256>>>(dout, r + 2); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-22s waves/SIMD=%d %8.3f ms %7.3f T lane-ops/s\n", name, bpc, best, (double)grid * 256 * ITERS * 32 / (best * 1e-3) / 1e12);
}
int main() {
  uint32_t* dout; CHECK(hipMalloc(&dout, 4096));
  for (int bpc : {2, 4}) {
    run<0>("v_ashrrev_i64", bpc, dout); run<1>("v_lshrrev_b64", bpc, dout); run<7>("v_lshlrev_b64", bpc, dout);
    run<2>("v_alignbit_b32", bpc, dout); run<3>("v_bfe_i32", bpc, dout); run<4>("v_and_b32 (literal)", bpc, dout);
    run<5>("v_mad_i64_i32 (x8)", bpc, dout); run<6>("v_lshl_add_u64", bpc, dout); run<8>("v_ashrrev_i32", bpc, dout);
    run<9>("v_mul_i32_i24", bpc, dout); run<10>("v_mul_hi_i32", bpc, dout); run<11>("v_add3_u32", bpc, dout); run<12>("v_lshl_add_u32", bpc, dout);
  }
  return 0;
}
