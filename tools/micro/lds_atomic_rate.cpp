// LDS atomic-add rate on gfx950: ds_add_u32 / ds_add_f32 / ds_add_f64 / ds_pk_add... per wave instruction, conflict-free addresses
// (lane i -> its own dword / qword), many waves per CU.   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics lds_atomic_rate.cpp -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int iters, int stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  // every thread its own element; `stride` (in elements) = 1: conflict-free; 0: all lanes of a wave on one address
  for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<double*>(smem)[i % 8192] = 0.0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    unsigned* p = reinterpret_cast<unsigned*>(smem) + (stride ? tid : (tid & ~63));
    for (int i = 0; i < iters; ++i) { atomicAdd(p + ((i & 7) << 10), 1u); }
  } else if (MODE == 1) {
    float* p = reinterpret_cast<float*>(smem) + (stride ? tid : (tid & ~63));
    for (int i = 0; i < iters; ++i) { unsafeAtomicAdd(p + ((i & 7) << 10), 1.5f); }
  } else if (MODE == 2) {
    double* p = reinterpret_cast<double*>(smem) + (stride ? tid : (tid & ~63));
    for (int i = 0; i < iters; ++i) { unsafeAtomicAdd(p + ((i & 7) << 10), 1.5); }
  } else {
    float* p = reinterpret_cast<float*>(smem) + (stride ? tid : (tid & ~63));
    for (int i = 0; i < iters; ++i) { atomicAdd(p + ((i & 7) << 10), 1.5f); }   // (safe form: whatever the compiler makes of it)
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int stride) {
  unsigned long long* d;
  hipMalloc(&d, 256 * 8);
  const int iters = 4096;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 131072, 0, d, iters, stride);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  const double ticks = s / 256;                       // s_memtime ticks (100 MHz constant clock on gfx950? report raw)
  const double per = ticks / ((double)iters * (threads / 64));
  printf("%-28s %4d threads  %s   %.3f ticks per wave instruction per CU\n", name, threads, stride ? "own element " : "one address ", per);
  hipFree(d);
}

int main() {
  for (int threads : {256, 1024}) {
    for (int stride : {1, 0}) {
      run<0>("ds_add_u32", threads, stride);
      run<1>("ds_add_f32 (unsafe)", threads, stride);
      run<2>("ds_add_f64 (unsafe)", threads, stride);
      run<3>("atomicAdd(float) plain", threads, stride);
    }
  }
  return 0;
}
