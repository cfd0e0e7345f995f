// What do the NON-VALU instructions of a wave cost on gfx950?  64 x v_pk_fma_f32 per loop body, alone and interleaved 1:1
// with s_nop / s_waitcnt / a scalar add / a DPP move, and in the window gather's mix; 1..4 waves per SIMD.  Reported: SIMD
// clocks per v_pk_fma_f32 (the unit pinned by "v_pk_fma_f32 alone at 4 waves per SIMD = 4 clocks").
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/issue_mix.bin tools/micro/issue_mix.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 256;
typedef float v2f __attribute__((ext_vector_type(2)));
#define B8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define B64(I) B8(I) B8(I) B8(I) B8(I) B8(I) B8(I) B8(I) B8(I)

template <int KIND>
__global__ void __launch_bounds__(1024) mix(unsigned long long* out, float seed) {
  float b = seed + threadIdx.x, c = seed * 0.5f;
  v2f p[8], q = {b, c}, r = {c, b};
  unsigned u[8], w = threadIdx.x * 3u + 1u, s = 7u;
  for (int i = 0; i < 8; ++i) { p[i] = v2f{b + i, c - i}; u[i] = w + i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) {
#define I(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q), "v"(r));
      B64(I)
#undef I
    } else if (KIND == 1) {
#define I(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\ts_nop 0" : "+v"(p[i]) : "v"(q), "v"(r));
      B64(I)
#undef I
    } else if (KIND == 2) {
#define I(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(p[i]) : "v"(q), "v"(r));
      B64(I)
#undef I
    } else if (KIND == 3) {
#define I(i) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\ts_add_u32 %1, %1, 3" : "+v"(p[i]), "+s"(s) : "v"(q), "v"(r) : "scc");
      B64(I)
#undef I
    } else if (KIND == 4) {
#define I(i) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_mov_b32_dpp %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(p[i]), "=v"(u[i]) : "v"(q), "v"(r), "v"(w));
      B64(I)
#undef I
    } else if (KIND == 5) {   // the gather's mix per half row: 4 pk_fma + dpp mov + s_nop 1 + s_waitcnt + v_add (address)
#define I(i) asm volatile("v_mov_b32_dpp %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\ts_waitcnt lgkmcnt(0)\n\t" \
                          "v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %0, %2, %3, %0\n\tv_add_u32_e32 %1, %4, %1" \
                          : "+v"(p[i]), "+v"(u[i]) : "v"(q), "v"(r), "v"(w));
      B8(I) B8(I)
#undef I
    } else if (KIND == 6) {   // the same without the scalar instructions
#define I(i) asm volatile("v_mov_b32_dpp %1, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t" \
                          "v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %0, %2, %3, %0\n\tv_add_u32_e32 %1, %4, %1" \
                          : "+v"(p[i]), "+v"(u[i]) : "v"(q), "v"(r), "v"(w));
      B8(I) B8(I)
#undef I
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f; unsigned x = s;
  for (int i = 0; i < 8; ++i) { acc += p[i].x + p[i].y; x ^= u[i]; }
  if (acc == 12345.678f && x == 77u) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static double g_unit = 1.0;   // ticks per SIMD clock
template <int KIND>
int run(const char* name, unsigned long long* dbuf, int cus, int fma_per_body) {
  printf("%-58s", name);
  for (int wps : {1, 2, 3, 4}) {
    const int threads = 256 * wps, waves = cus * 4 * wps;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(mix<KIND>, dim3(cus), dim3(threads), 0, 0, dbuf, 1.0f);
      CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(1 + waves);
    CHECK(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin() + 1, h.end());
    const double per_fma_per_wave = (double)h[1 + waves / 2] / ((double)ITER * fma_per_body);
    if (KIND == 0 && wps == 4) g_unit = per_fma_per_wave / 4 / 4.0;
    printf("  %dw: %6.2f", wps, per_fma_per_wave / wps);     // ticks per fma per SIMD
  }
  printf("   (ticks per v_pk_fma_f32 per SIMD)\n");
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  unsigned long long* dbuf;
  CHECK(hipMalloc(&dbuf, (1 + 256 * 16) * 8 + 1024));
  printf("%s, %d CUs; columns: waves per SIMD\n", p.gcnArchName, cus);
  run<0>("v_pk_fma_f32 alone", dbuf, cus, 64);
  run<1>("v_pk_fma_f32 + s_nop 0", dbuf, cus, 64);
  run<2>("v_pk_fma_f32 + s_waitcnt lgkmcnt(0)", dbuf, cus, 64);
  run<3>("v_pk_fma_f32 + s_add_u32", dbuf, cus, 64);
  run<4>("v_pk_fma_f32 + v_mov_b32_dpp", dbuf, cus, 64);
  run<5>("gather mix: dpp, s_nop 1, s_waitcnt, 4 pk_fma, v_add", dbuf, cus, 64);
  run<6>("gather mix without the two scalar instructions", dbuf, cus, 64);
  printf("unit: v_pk_fma_f32 alone at 4 waves per SIMD = 4 clocks -> %.3f ticks per clock\n", g_unit);
  return 0;
}
