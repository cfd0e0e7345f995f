// VALU issue cost per wave64 instruction on gfx950, by instruction kind and waves per SIMD (the question behind
// "26.5 M VALU instructions = 49 us": does every VALU instruction hold its SIMD for 4 clocks?).
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/valu_rate.bin tools/micro/valu_rate.cpp && tools/micro/valu_rate.bin
// Each wave runs ITER x 64 instructions of one kind on 8 independent register chains between two s_memtime reads.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 256;

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BODY64(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)

template <int KIND>
__global__ void __launch_bounds__(1024) rate(unsigned long long* out, float seed) {
  float a[8], b = seed + threadIdx.x, c = seed * 0.5f;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[8], q = {b, c}, r = {c, b};
  unsigned u[8], w = threadIdx.x * 3u + 1u;
  for (int i = 0; i < 8; ++i) { a[i] = b + i; p[i] = v2f{b + i, c - i}; u[i] = w + i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) {
#define INS(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      BODY64(INS)
#undef INS
    } else if (KIND == 1) {
#define INS(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q), "v"(r));
      BODY64(INS)
#undef INS
    } else if (KIND == 2) {
#define INS(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(w));
      BODY64(INS)
#undef INS
    } else if (KIND == 3) {
#define INS(i) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(w));
      BODY64(INS)
#undef INS
    } else if (KIND == 4) {
#define INS(i) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(w));
      BODY64(INS)
#undef INS
    } else if (KIND == 5) {
#define INS(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(u[i]) : "v"(w), "v"(w));
      BODY64(INS)
#undef INS
    } else if (KIND == 6) {
#define INS(i) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(w));
      BODY64(INS)
#undef INS
    } else if (KIND == 7) {
#define INS(i) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(u[i]) : "v"(w) : "vcc");
      BODY64(INS)
#undef INS
    } else if (KIND == 8) {
#define INS(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u[i]) : "v"(w));
      BODY64(INS)
#undef INS
    } else if (KIND == 9) {
#define INS(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(q));
      BODY64(INS)
#undef INS
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f; unsigned x = 0;
  for (int i = 0; i < 8; ++i) { s += a[i] + p[i].x + p[i].y; x ^= u[i]; }
  if (s == 12345.678f && x == 77u) out[0] = 1;   // keeps the chains alive
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
int run(const char* name, unsigned long long* dbuf, int cus) {
  for (int wps : {1, 2, 4}) {                       // waves per SIMD: one workgroup of 4 * wps waves per CU
    const int threads = 256 * wps, waves = cus * 4 * wps;
    hipLaunchKernelGGL(rate<KIND>, dim3(cus), dim3(threads), 0, 0, dbuf, 1.0f);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(rate<KIND>, dim3(cus), dim3(threads), 0, 0, dbuf, 1.0f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(1 + waves);
    CHECK(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin() + 1, h.end());
    const double n = (double)ITER * 64;
    printf("%-16s %d wave/SIMD: median %8.3f ticks per instruction per wave (min %.3f, max %.3f)\n", name, wps,
           h[1 + waves / 2] / n, h[1] / n, h[waves] / n);
  }
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("%s: %d CUs, clockRate %d kHz; ticks are s_memtime units (compare kinds: the ratios are what matters)\n", p.gcnArchName, cus, p.clockRate);
  unsigned long long* dbuf;
  CHECK(hipMalloc(&dbuf, (1 + 256 * 16) * 8 + 1024));
  run<0>("v_fma_f32", dbuf, cus);
  run<1>("v_pk_fma_f32", dbuf, cus);
  run<9>("v_pk_mul_f32", dbuf, cus);
  run<2>("v_mov_b32_dpp", dbuf, cus);
  run<6>("v_add_u32_dpp", dbuf, cus);
  run<3>("v_add_u32", dbuf, cus);
  run<4>("v_xor_b32", dbuf, cus);
  run<5>("v_mad_u32_u24", dbuf, cus);
  run<7>("v_cndmask_b32", dbuf, cus);
  run<8>("v_mul_lo_u32", dbuf, cus);
  // wall-clock cross-check of the tick unit: one long launch timed with events
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(rate<1>, dim3(cus), dim3(1024), 0, 0, dbuf, 1.0f);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(1 + cus * 16);
  CHECK(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin() + 1, h.end());
  printf("v_pk_fma_f32 4 waves/SIMD: %.1f us per launch by events; median wave %llu ticks -> %.1f ticks/us; %d x 64 instr x 4 waves per SIMD -> %.3f ns per instruction per SIMD\n",
         ms * 1e3 / 20, h[1 + cus * 8], h[1 + cus * 8] / (ms * 1e3 / 20), ITER, ms * 1e6 / 20 / (ITER * 64.0 * 4));
  return 0;
}
