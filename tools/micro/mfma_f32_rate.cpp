// Micro-benchmark: issue interval and dependent-accumulator latency of the fp32-input MFMA forms of gfx950 next to
// v_pk_fma_f32 -- the numbers behind the kernel choice of the dynamic mask head (uninext_amd/csrc/dynmask.hip).
// One 256-thread workgroup per CU x 4 (4 waves per SIMD) or x 1; every wave runs `iters` rounds of CH independent
// accumulator chains.  Reports cycles per instruction per SIMD (wall time x 2.4 GHz / instructions per SIMD) and the
// resulting MAC rate.    hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.cpp -o mfma_f32_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FORM, int CH>
__global__ void __launch_bounds__(256) k(int iters, float* out) {
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  float res = 0.f;
  if constexpr (FORM == 0) {          // v_mfma_f32_4x4x1_16b_f32
    f32x4 c[CH];
    for (int i = 0; i < CH; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < CH; ++i) res += c[i][0];
  } else if constexpr (FORM == 1) {   // v_mfma_f32_16x16x1_4b_f32
    f32x16 c[CH];
    for (int i = 0; i < CH; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < CH; ++i) res += c[i][0];
  } else if constexpr (FORM == 2) {   // v_mfma_f32_16x16x4_f32
    f32x4 c[CH];
    for (int i = 0; i < CH; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < CH; ++i) res += c[i][0];
  } else if constexpr (FORM == 3) {   // v_mfma_f32_32x32x2_f32
    f32x16 c[CH];
    for (int i = 0; i < CH; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < CH; ++i) res += c[i][0];
  } else {                            // v_pk_fma_f32
    f32x2 c[CH];
    const f32x2 av{a, a + 1.f}, bv{b, b};
    for (int i = 0; i < CH; ++i) c[i] = f32x2{0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < CH; ++i) { c[i] = __builtin_elementwise_fma(av, bv, c[i]); asm volatile("" : "+v"(c[i])); }
    for (int i = 0; i < CH; ++i) res += c[i][0];
  }
  if (res == 123.456f) out[0] = res;
}

template <int FORM, int CH>
void run(const char* name, double macs_per_instr, int wg_per_cu) {
  float* out;
  (void)hipMalloc(&out, 4);
  const int iters = 20000 / CH * 8 / 8;
  const int cus = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FORM, CH>), dim3(cus * wg_per_cu), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * CH * wg_per_cu;      // one wave of every workgroup per SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
  printf("%-28s chains/wave %2d  waves/SIMD %d  %7.2f cyc/instr/SIMD  %6.1f MAC/clk/SIMD  (%.1f TFLOP/s chip at 2.4 GHz)\n", name, CH,
         wg_per_cu, cyc, macs_per_instr / cyc, 2.0 * macs_per_instr / cyc * 1024 * 2.4e9 / 1e12);
  (void)hipFree(out);
}

int main() {
  for (int w : {1, 4}) {
    run<0, 1>("v_mfma_f32_4x4x1_16b_f32", 256, w);
    run<0, 4>("v_mfma_f32_4x4x1_16b_f32", 256, w);
    run<0, 8>("v_mfma_f32_4x4x1_16b_f32", 256, w);
    run<1, 1>("v_mfma_f32_16x16x1_4b_f32", 1024, w);
    run<1, 4>("v_mfma_f32_16x16x1_4b_f32", 1024, w);
    run<2, 1>("v_mfma_f32_16x16x4_f32", 1024, w);
    run<2, 4>("v_mfma_f32_16x16x4_f32", 1024, w);
    run<3, 1>("v_mfma_f32_32x32x2_f32", 2048, w);
    run<3, 2>("v_mfma_f32_32x32x2_f32", 2048, w);
    run<4, 1>("v_pk_fma_f32", 128, w);
    run<4, 8>("v_pk_fma_f32", 128, w);
  }
  return 0;
}
