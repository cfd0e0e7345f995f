// Do LDS reads and vector ALU work overlap on a gfx950 SIMD?  (round 6: the question every LDS-window formulation of the
// MSDeformAttn forward comes down to -- the gather is 16 ds_read_b128 + 32 v_pk_fma_f32 per (pair, sample) whatever the lane layout.)
// One "unit" = 1 ds_read_b128 (conflict-free: lane-linear addresses) and / or 2 v_pk_fma_f32; 16 units per loop body; 1..4 waves
// per SIMD, every CU busy.  Reported: SIMD clocks per unit (the clock is pinned by "VALU only at 4 waves per SIMD = 8 clocks").
//   KIND 0  LDS only            16 reads, s_waitcnt lgkmcnt(0) once per body
//   KIND 1  VALU only           32 packed FMAs on 8 independent accumulators
//   KIND 2  both, independent   read + 2 FMAs per unit, the FMAs do not touch the read data; one wait per body
//   KIND 3  both, dependent     the window pass's shape: the FMAs of unit u consume the read of unit u - 8 (ring of 8, lgkmcnt(7))
//   KIND 4  both, dependent, ring of 16 (lgkmcnt(15))
//   KIND 5  as 3 with 4 v_fma_f32 instead of 2 v_pk_fma_f32
//   KIND 6  LDS only, ds_read_b64 x 2 per unit (same bytes)
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/lds_valu_overlap.bin tools/micro/lds_valu_overlap.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 256;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((__vector_size__(16)));
#define B8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define B16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)

template <int KIND>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, float seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int o = threadIdx.x * 16; o < 64 * 1024; o += blockDim.x * 16) *reinterpret_cast<f32x4*>(smem + o) = f32x4{seed, 1.f, 2.f, 3.f};
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + lane * 16u + (wave & 3u) * 16384u;
  float b = seed + threadIdx.x, c = seed * 0.5f;
  v2f p[8], q = {b, c}, r = {c, b};
  f32x4 d[16];
  for (int i = 0; i < 8; ++i) p[i] = v2f{b + i, c - i};
  for (int i = 0; i < 16; ++i) d[i] = f32x4{b, c, b, c};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) {
#define I(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[i]) : "v"(addr), "i"(1024 * (i)));
      B16(I)
#undef I
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (KIND == 1) {
#define I(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_fma_f32 %3, %1, %2, %3" : "+v"(p[(i) & 7]), "+v"(q) : "v"(r), "v"(p[((i) + 4) & 7]));
#undef I
#define I(i) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %2, %3, %1" : "+v"(p[(i) & 7]), "+v"(p[((i) + 4) & 7]) : "v"(q), "v"(r));
      B16(I)
#undef I
    } else if (KIND == 2) {
#define I(i) asm volatile("ds_read_b128 %2, %5 offset:%6\n\tv_pk_fma_f32 %0, %3, %4, %0\n\tv_pk_fma_f32 %1, %3, %4, %1" \
                          : "+v"(p[(i) & 7]), "+v"(p[((i) + 4) & 7]), "=v"(d[i]) : "v"(q), "v"(r), "v"(addr), "i"(1024 * (i)));
      B16(I)
#undef I
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (KIND == 3 || KIND == 4 || KIND == 5) {
      // the window pass's shape: the FMAs of unit i use ring register i % RING (requested RING units ago), then it is re-requested
      constexpr int RING = KIND == 4 ? 16 : 8;
      typedef const f32x4 __attribute__((address_space(3)))* lds4;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x4 v = d[i % RING];
        if (KIND == 5) {
          p[i & 7].x = __builtin_fmaf(b, v[0], p[i & 7].x); p[i & 7].y = __builtin_fmaf(b, v[1], p[i & 7].y);
          p[(i + 4) & 7].x = __builtin_fmaf(b, v[2], p[(i + 4) & 7].x); p[(i + 4) & 7].y = __builtin_fmaf(b, v[3], p[(i + 4) & 7].y);
        } else {
          p[i & 7] = __builtin_elementwise_fma(q, v2f{v[0], v[1]}, p[i & 7]);
          p[(i + 4) & 7] = __builtin_elementwise_fma(q, v2f{v[2], v[3]}, p[(i + 4) & 7]);
        }
        asm volatile("" : "+v"(p[i & 7]), "+v"(p[(i + 4) & 7]));
        d[i % RING] = reinterpret_cast<lds4>((unsigned long long)addr)[64 * i];
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (KIND == 6) {
#define I(i) asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=v"(p[(i) & 7]), "=v"(p[((i) + 4) & 7]) : "v"(addr), "i"(1024 * (i)), "i"(1024 * (i) + 512));
      B16(I)
#undef I
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y;
  for (int i = 0; i < 16; ++i) acc += d[i][0] + d[i][3];
  if (acc == 12345.678f) out[0] = 1;
  if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static double g_unit = 1.0;   // ticks per SIMD clock
template <int KIND>
int run(const char* name, unsigned long long* dbuf, int cus) {
  printf("%-62s", name);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  for (int wps : {1, 2, 3, 4}) {
    const int threads = 256 * wps, waves = cus * 4 * wps;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k<KIND>, dim3(cus), dim3(threads), 64 * 1024, 0, dbuf, 1.0f);
      CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h(1 + waves);
    CHECK(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin() + 1, h.end());
    const double per_unit_per_simd = (double)h[1 + waves / 2] / ((double)ITER * 16) / wps;   // ticks per unit per SIMD
    if (KIND == 1 && wps == 4) g_unit = per_unit_per_simd / 8.0;
    printf("  %dw: %6.2f", wps, per_unit_per_simd);
  }
  printf("   (ticks per unit per SIMD)\n");
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* dbuf;
  CHECK(hipMalloc(&dbuf, (1 + cus * 16) * 8));
  printf("%s: %d CUs; unit = 1 ds_read_b128 and / or 2 v_pk_fma_f32; columns: waves per SIMD\n", prop.gcnArchName, cus);
  if (run<1>("VALU only (2 v_pk_fma_f32)", dbuf, cus)) return 1;
  if (run<0>("LDS only (1 ds_read_b128, wait once per 16)", dbuf, cus)) return 1;
  if (run<6>("LDS only (2 ds_read_b64, wait once per 16)", dbuf, cus)) return 1;
  if (run<2>("both, FMAs independent of the reads", dbuf, cus)) return 1;
  if (run<3>("both, FMAs consume the read of 8 units ago (lgkmcnt 7)", dbuf, cus)) return 1;
  if (run<4>("both, FMAs consume the read of 16 units ago (lgkmcnt 15)", dbuf, cus)) return 1;
  if (run<5>("both, 4 v_fma_f32 consume the read of 8 units ago", dbuf, cus)) return 1;
  printf("clock unit: %.3f ticks per SIMD clock (VALU only at 4 waves per SIMD = 8 clocks per unit)\n", g_unit);
  printf("in SIMD clocks per unit at 4 waves per SIMD: divide the last column by that\n");
  return 0;
}
