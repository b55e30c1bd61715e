// Integer-VALU peak microbenchmark for gfx950 (MI355X).
// Measures sustained issue rate of the instructions the Montgomery field
// arithmetic is built from, so the "integer roofline" is a measured number.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 32768;
constexpr int UNROLL = 16;  // instructions per loop iteration per accumulator set

template <int MODE>
__global__ void __launch_bounds__(256) k_bench(uint32_t* out, uint32_t seed) {
  uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
  uint64_t acc[8];
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { acc[i] = a + i; d[i] = 1.0 + i + a; }
  double da = 1.0000001 + a, db = 0.999999;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MODE == 0) {  // v_mad_u64_u32
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 1) {  // v_mul_lo_u32
          uint32_t lo = (uint32_t)acc[i];
          asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(lo) : "v"(a));
          acc[i] = lo;
        } else if (MODE == 2) {  // v_mul_hi_u32
          uint32_t lo = (uint32_t)acc[i];
          asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(lo) : "v"(a));
          acc[i] = lo;
        } else if (MODE == 3) {  // v_fma_f64
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
        } else if (MODE == 4) {  // v_add_co_u32 (full-rate int add with carry out)
          uint32_t lo = (uint32_t)acc[i];
          asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(lo) : "v"(a) : "vcc");
          acc[i] = lo;
        } else if (MODE == 5) {  // v_mad_u32_u24
          uint32_t lo = (uint32_t)acc[i];
          asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b));
          acc[i] = lo;
        } else if (MODE == 6) {  // v_lshl_add_u64
          asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
        } else if (MODE == 7) {  // v_addc_co_u32 chain
          uint32_t lo = (uint32_t)acc[i];
          asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(lo) : "v"(a) : "vcc");
          acc[i] = lo;
        } else if (MODE == 8) {  // v_fma_f32
          float f = __uint_as_float((uint32_t)acc[i]);
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f) : "v"(__uint_as_float(a)), "v"(__uint_as_float(b)));
          acc[i] = __float_as_uint(f);
        } else if (MODE == 9) {  // v_mad_u64_u32 with sgpr carry-out, separate dst
          asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "s10", "s11");
        }
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i] + (uint64_t)d[i];
  if (s == 0x1234567) out[0] = (uint32_t)s;
}

template <int MODE>
double run(const char* name, int blocks_per_cu, uint32_t* dout) {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int grid = prop.multiProcessorCount * blocks_per_cu;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k_bench<MODE><<<grid, 256>>>(dout, 1); CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < 5; r++) {
    CHECK(hipEventRecord(e0));
    k_bench<MODE><<<grid, 256>>>(dout, r + 2);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  double ops = (double)grid * 256 * ITERS * UNROLL;
  double rate = ops / (best * 1e-3);
  // cycles per wave-instruction per SIMD, assuming clk from prop
  double clk = prop.clockRate * 1e3;
  double simds = prop.multiProcessorCount * 4.0;
  double wave_instr = ops / 64.0;
  double cyc = (best * 1e-3) * clk * simds / wave_instr;
  printf("%-28s blocks/CU=%d  %8.3f ms  %8.3f Tops/s  %.2f cyc/wave-instr/SIMD (at %.0f MHz)\n",
         name, blocks_per_cu, best, rate / 1e12, cyc, clk / 1e6);
  return rate;
}

int main() {
  uint32_t* dout; CHECK(hipMalloc(&dout, 4096));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  for (int bpc : {1, 4, 8}) {
    run<0>("v_mad_u64_u32 (vcc)", bpc, dout);
    run<9>("v_mad_u64_u32 (sgpr cout)", bpc, dout);
    run<1>("v_mul_lo_u32", bpc, dout);
    run<2>("v_mul_hi_u32", bpc, dout);
    run<3>("v_fma_f64", bpc, dout);
    run<4>("v_add_co_u32", bpc, dout);
    run<7>("v_addc_co_u32", bpc, dout);
    run<5>("v_mad_u32_u24", bpc, dout);
    run<6>("v_lshl_add_u64", bpc, dout);
    run<8>("v_fma_f32", bpc, dout);
  }
  return 0;
}
