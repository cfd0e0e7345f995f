// Does buffer_load_dwordx4 ... lds (LDS-DMA) reach the whole 160 KB of a gfx950 workgroup's LDS?  Each chunk of 1 KB is
// fetched from global memory into LDS offset `off` by DMA, read back with ds_read and compared.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/lds_dma_range.bin tools/micro/lds_dma_range.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((__vector_size__(16)));

__global__ void __launch_bounds__(64) k(const float* src, int* bad, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  for (int o = lane * 16; o < nchunks * 1024; o += 64 * 16) *reinterpret_cast<f32x4*>(smem + o) = f32x4{-1.f, -1.f, -1.f, -1.f};
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nchunks * 1024, 0x00020000);
  for (int c = 0; c < nchunks; ++c)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, (unsigned)(lane * 16 + c * 1024), 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(smem + c * 1024 + lane * 16);
    const float want = (float)(c * 256 + lane * 4);
    const bool ok = v[0] == want && v[1] == want + 1 && v[2] == want + 2 && v[3] == want + 3;
    if (!ok) atomicAdd(&bad[c], 1);
  }
}

int main() {
  const int nchunks = 159;
  std::vector<float> h(nchunks * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  float* src; int* bad;
  CHECK(hipMalloc(&src, h.size() * 4)); CHECK(hipMalloc(&bad, nchunks * 4));
  CHECK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(bad, 0, nchunks * 4));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, nchunks * 1024));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), nchunks * 1024, 0, src, bad, nchunks);
  CHECK(hipDeviceSynchronize());
  std::vector<int> hb(nchunks);
  CHECK(hipMemcpy(hb.data(), bad, nchunks * 4, hipMemcpyDeviceToHost));
  int first_bad = -1, nbad = 0;
  for (int c = 0; c < nchunks; ++c) if (hb[c]) { if (first_bad < 0) first_bad = c; ++nbad; }
  printf("LDS-DMA into %d KB of LDS: %d chunks wrong, first wrong chunk at %d KB\n", nchunks, nbad, first_bad);
  return 0;
}
