// Micro-benchmark: how fast can a 512-thread workgroup pull 74 KB of scattered 128-byte lines (a pixel window of one
// head: 1 KB pixel stride) into LDS on gfx950 -- LDS-DMA (buffer_load_dwordx4 ... lds) vs plain buffer loads into
// registers followed by ds_write_b128 -- with 1 or 2 workgroups per CU, from an L2-resident or an HBM-sized footprint.
// hipcc --offload-arch=gfx950 -O3 lds_stage_bench.cpp -o lds_stage_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
constexpr int kChunks = 74;   // 1 KB chunks per window set (592 slots)

template <int MODE>   // 0: LDS-DMA, 1: registers + ds_write_b128, 2: registers only (no LDS write)
__global__ void __launch_bounds__(512, 4) k(const float* __restrict__ base, long long pixels, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(pixels * 1024), 0x00020000);
  const uint32_t hoff = (blockIdx.x & 7) * 128u;
  uint32_t seed = blockIdx.x * 9781u + 12345u;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    seed = seed * 1664525u + 1013904223u;
    const uint32_t pix0 = (seed >> 8) % (uint32_t)(pixels - 4096);     // window origin: a pseudo-random pixel
    u32x4 r[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const int i = wv + 8 * t;
      if (i < kChunks) {
        // 8 slots of a chunk: consecutive pixels of a row, rows 167 pixels apart (level-0 like)
        const uint32_t slot = 8 * i + (lane >> 3);
        const uint32_t pix = pix0 + (slot / 22) * 167u + (slot % 22);
        const uint32_t off = pix * 1024u + (lane & 7) * 16u;
        if (MODE == 0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, off, hoff, 0, 0);
        else
          r[t] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, hoff, 0);
      }
    }
    if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 10; ++t) {
        const int i = wv + 8 * t;
        if (i < kChunks) *reinterpret_cast<u32x4*>(smem + i * 1024 + lane * 16) = r[t];
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < 10; ++t) if (wv + 8 * t < kChunks) acc += __builtin_bit_cast(float, r[t][0]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += reinterpret_cast<float*>(smem)[(threadIdx.x * 37 + it) & 8191];
    __syncthreads();
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const float* buf, long long pixels, int wgs, float* out) {
  const int iters = 40;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 78 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(512), 78 * 1024, 0, buf, pixels, 2, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(512), 78 * 1024, 0, buf, pixels, iters, out);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * iters * kChunks * 1024.0;
  printf("%-28s %4d WGs  footprint %6.1f MB: %7.3f ms  %7.1f GB/s chip  %6.1f GB/s per CU  %5.2f us per 74 KB window set\n", name, wgs,
         pixels * 1024.0 / 1e6, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 256, ms * 1e3 / iters);
}

int main() {
  const long long big = 400000, small = 24000;      // 410 MB (HBM) and 24.6 MB (L2 / MALL resident)
  float* buf; float* out;
  hipMalloc(&buf, big * 1024); hipMalloc(&out, 4096 * 512 * 4);
  hipMemset(buf, 0, big * 1024);
  for (long long px : {small, big})
    for (int wgs : {256, 512}) {
      run<0>("LDS-DMA", buf, px, wgs, out);
      run<1>("registers + ds_write_b128", buf, px, wgs, out);
      run<2>("registers only", buf, px, wgs, out);
    }
  return 0;
}
