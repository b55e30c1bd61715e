// Co-residency microbenchmark for gfx950: does a light kernel on a second stream run WHILE a register-heavy
// VALU-bound kernel (the shape of k_accum: 2 waves per SIMD at ~224 VGPRs each) fills the chip, and what does
// each lose?  Decides whether the MSM's sort of later windows can hide under the accumulation of earlier ones.
//   P: 256-thread workgroups, 2 per CU, ~104 live 64-bit accumulators per lane, multiply-adds only.
//   Q: a streaming copy with a per-workgroup LDS array, workgroups of 256 or 1024 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int NACC = 104;

__global__ void __launch_bounds__(256, 2) k_heavy(uint32_t* out, uint32_t seed, int iters) {
  __shared__ uint32_t plds[768];  // k_accum holds 3 KB
  plds[threadIdx.x] = seed;
  __syncthreads();
  int32_t a = (int32_t)(seed + threadIdx.x + (plds[(threadIdx.x * 3u) & 255u] & 1u)), b = (int32_t)(seed * 3 + blockIdx.x);
  int64_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = a + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < NACC; u++)
      asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[u]) : "v"(a), "v"(b) : "vcc");
  }
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i];
  if (s == 0x1234567) out[0] = (uint32_t)s;
}

template <int BLOCK, int LDS_WORDS>
__global__ void __launch_bounds__(BLOCK) k_light(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  __shared__ uint32_t lds[LDS_WORDS];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const uint32_t salt = lds[(threadIdx.x * 7u) % BLOCK] & 1u;  // keeps the array allocated
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint4 v = src[i];
    v.x += salt;
    dst[i] = v;
  }
}

// a light kernel that is a chain of plain VALU work (what a sort kernel's index arithmetic looks like to the issue
// arbiter): does a third wave on a SIMD get issue slots beside two saturating ones, and does s_setprio change that?
template <int PRIO>
__global__ void __launch_bounds__(256, 8) k_valu_light(uint32_t* out, uint32_t seed, int iters) {
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  uint32_t a = seed + threadIdx.x, b = seed * 7u + blockIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  }
  if (a == 0x1234567u) out[1] = a;
}

// the same with NR live registers per lane: does a wave that needs most of the 64 registers k_accum's two waves
// leave on a SIMD (512 - 2 x 224) find them in one piece?
template <int NR, int LDS_WORDS = 64>
__global__ void __launch_bounds__(256, 8) k_valu_regs(uint32_t* out, uint32_t seed, int iters) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ uint32_t lds[LDS_WORDS];
  lds[threadIdx.x % LDS_WORDS] = seed;
  __syncthreads();
  seed += lds[(threadIdx.x * 5u) % LDS_WORDS] & 1u;
  uint32_t a[NR];
#pragma unroll
  for (int i = 0; i < NR; i++) a[i] = seed + threadIdx.x + i;
  const uint32_t b = seed * 7u + blockIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < NR; u++) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[u]) : "v"(b));
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < NR; i++) s += a[i];
  if (s == 0x1234567u) out[1] = s;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t bytes = 512ull << 20, n = bytes / 16;
  uint4 *src, *dst;
  uint32_t* dout;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMalloc(&dst, bytes));
  CHECK(hipMalloc(&dout, 64));
  CHECK(hipMemset(src, 1, bytes));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, eP, eQ;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&eP));
  CHECK(hipEventCreate(&eQ));
  const int iters = 20000;  // about a millisecond of multiply-adds

  auto launchQ = [&](int variant, hipStream_t st) {
    switch (variant) {
      case 0: k_light<256, 6 * 1024><<<cus * 4, 256, 0, st>>>(src, dst, n); break;    // 24 KB of LDS
      case 1: k_light<1024, 24 * 1024><<<cus, 1024, 0, st>>>(src, dst, n); break;     // 96 KB of LDS
      case 2: k_light<256, 256><<<cus * 8, 256, 0, st>>>(src, dst, n); break;         // no LDS to speak of
    }
  };
  const char* qname[3] = {"Q 256 threads, 24 KB LDS", "Q 1024 threads, 96 KB LDS", "Q 256 threads, 1 KB LDS"};

  // warm-up
  k_heavy<<<cus * 2, 256, 0, s1>>>(dout, 1, 100);
  for (int v = 0; v < 3; v++) launchQ(v, s2);
  CHECK(hipDeviceSynchronize());

  float tP;
  CHECK(hipEventRecord(e0, s1));
  k_heavy<<<cus * 2, 256, 0, s1>>>(dout, 2, iters);
  CHECK(hipEventRecord(eP, s1));
  CHECK(hipEventSynchronize(eP));
  CHECK(hipEventElapsedTime(&tP, e0, eP));
  printf("P alone (2 waves/SIMD, %d accumulators): %.3f ms\n", NACC, tP);

  for (int v = 0; v < 3; v++) {
    float tQ, tP2, tQ2;
    CHECK(hipEventRecord(e0, s2));
    launchQ(v, s2);
    CHECK(hipEventRecord(eQ, s2));
    CHECK(hipEventSynchronize(eQ));
    CHECK(hipEventElapsedTime(&tQ, e0, eQ));
    // both: P first on s1, Q on s2 right behind the common start
    CHECK(hipEventRecord(e0, s1));
    CHECK(hipStreamWaitEvent(s2, e0, 0));
    k_heavy<<<cus * 2, 256, 0, s1>>>(dout, 3 + v, iters);
    CHECK(hipEventRecord(eP, s1));
    launchQ(v, s2);
    CHECK(hipEventRecord(eQ, s2));
    CHECK(hipEventSynchronize(eP));
    CHECK(hipEventSynchronize(eQ));
    CHECK(hipEventElapsedTime(&tP2, e0, eP));
    CHECK(hipEventElapsedTime(&tQ2, e0, eQ));
    printf("%-28s alone %.3f ms (%.2f TB/s) | with P: Q done at %.3f ms, P done at %.3f ms (P alone %.3f)\n", qname[v], tQ,
           2.0 * bytes / (tQ * 1e-3) / 1e12, tQ2, tP2, tP);
  }
  for (int prio = 0; prio < 2; prio++) {
    const int li = 4000;  // 64 K dependent VALU instructions per wave
    float tQ, tP2, tQ2;
    auto launch = [&](hipStream_t st) {
      if (prio) k_valu_light<3><<<cus, 256, 0, st>>>(dout, 5, li); else k_valu_light<0><<<cus, 256, 0, st>>>(dout, 5, li);
    };
    launch(s2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, s2));
    launch(s2);
    CHECK(hipEventRecord(eQ, s2));
    CHECK(hipEventSynchronize(eQ));
    CHECK(hipEventElapsedTime(&tQ, e0, eQ));
    CHECK(hipEventRecord(e0, s1));
    CHECK(hipStreamWaitEvent(s2, e0, 0));
    k_heavy<<<cus * 2, 256, 0, s1>>>(dout, 9, iters / 8);
    CHECK(hipEventRecord(eP, s1));
    launch(s2);
    CHECK(hipEventRecord(eQ, s2));
    CHECK(hipEventSynchronize(eP));
    CHECK(hipEventSynchronize(eQ));
    CHECK(hipEventElapsedTime(&tP2, e0, eP));
    CHECK(hipEventElapsedTime(&tQ2, e0, eQ));
    printf("VALU chain, one wave per SIMD, s_setprio %d: alone %.3f ms | beside P: done at %.3f ms (P done at %.3f ms, alone %.3f)\n",
           prio ? 3 : 0, tQ, tQ2, tP2, tP / 8);
  }
  for (int v = 0; v < 11; v++) {
    float tQ, tP2, tQ2;
    auto launch = [&](hipStream_t st) {
      if (v == 3) k_valu_regs<50, 18 * 1024><<<cus, 256, 0, st>>>(dout, 5, 64000 / 50);       // 72 KB of LDS
      if (v == 4) k_valu_regs<50, 18 * 1024><<<cus * 5, 256, 0, st>>>(dout, 5, 64000 / 50);   // five rounds of workgroups
      if (v == 5) k_valu_regs<34, 26 * 1024><<<cus * 5, 256, 0, st>>>(dout, 5, 64000 / 34);   // 104 KB of LDS
      if (v == 6) k_valu_regs<50, 2 * 1024><<<cus, 256, 0, st>>>(dout, 5, 64000 / 50);
      if (v == 7) k_valu_regs<50, 4 * 1024><<<cus, 256, 0, st>>>(dout, 5, 64000 / 50);
      if (v == 8) k_valu_regs<50, 8 * 1024><<<cus, 256, 0, st>>>(dout, 5, 64000 / 50);
      if (v == 9) k_valu_regs<50, 12 * 1024><<<cus, 256, 0, st>>>(dout, 5, 64000 / 50);
      if (v == 10) k_valu_regs<50, 14 * 1024><<<cus, 256, 0, st>>>(dout, 5, 64000 / 50);
      if (v == 0) k_valu_regs<20><<<cus, 256, 0, st>>>(dout, 5, 64000 / 20);
      if (v == 1) k_valu_regs<34><<<cus, 256, 0, st>>>(dout, 5, 64000 / 34);
      if (v == 2) k_valu_regs<50><<<cus, 256, 0, st>>>(dout, 5, 64000 / 50);
    };
    launch(s2);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, s2));
    launch(s2);
    CHECK(hipEventRecord(eQ, s2));
    CHECK(hipEventSynchronize(eQ));
    CHECK(hipEventElapsedTime(&tQ, e0, eQ));
    CHECK(hipEventRecord(e0, s1));
    CHECK(hipStreamWaitEvent(s2, e0, 0));
    k_heavy<<<cus * 2, 256, 0, s1>>>(dout, 9, iters / 8);
    CHECK(hipEventRecord(eP, s1));
    launch(s2);
    CHECK(hipEventRecord(eQ, s2));
    CHECK(hipEventSynchronize(eP));
    CHECK(hipEventSynchronize(eQ));
    CHECK(hipEventElapsedTime(&tP2, e0, eP));
    CHECK(hipEventElapsedTime(&tQ2, e0, eQ));
    static const char* names[11] = {"20 registers", "34 registers", "50 registers", "50 registers, 72 KB LDS",
                                    "50 registers, 72 KB LDS, 5 workgroups per CU", "34 registers, 104 KB LDS, 5 workgroups per CU",
                                    "50 registers, 8 KB LDS", "50 registers, 16 KB LDS", "50 registers, 32 KB LDS",
                                    "50 registers, 48 KB LDS", "50 registers, 56 KB LDS"};
    printf("VALU work, %s, s_setprio 3: alone %.3f ms | beside P: done at %.3f ms (P done at %.3f ms)\n", names[v], tQ, tQ2, tP2);
  }
  return 0;
}
