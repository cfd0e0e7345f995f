// Micro-benchmark: throughput of LDS atomic adds on gfx950 (one 256-thread workgroup per CU slot, conflict-free
// lane-distinct addresses).  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_bench.cpp -o lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T, int MODE>
__global__ void k(T* out, int iters) {
  __shared__ T buf[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = 0;
  __syncthreads();
  T v = (T)(threadIdx.x + 1);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = (threadIdx.x + 256 * u) & 4095;
      if (MODE == 0) __hip_atomic_fetch_add(&buf[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else buf[idx] += v;   // plain read-modify-write (racy across waves, rate reference only)
    }
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x];
}

template <typename T, int MODE>
void run(const char* name) {
  T* out;
  hipMalloc(&out, sizeof(T) * 1024 * 256);
  const int iters = 200;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<T, MODE>), dim3(1024), dim3(256), 0, 0, out, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<T, MODE>), dim3(1024), dim3(256), 0, 0, out, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double wave_instr = 1024.0 * 4 * iters * 16;          // wave64 instructions
  const double per_cu_cycles = ms * 1e-3 * 2.4e9 / (wave_instr / 256);
  printf("%-18s %8.3f ms  %6.1f G lane-ops/s  ~%5.1f cycles per wave instruction per CU (at 2.4 GHz)\n", name, ms,
         wave_instr * 64 / ms / 1e6, per_cu_cycles);
  hipFree(out);
}

int main() {
  run<float, 0>("ds_add_f32");
  run<unsigned, 0>("ds_add_u32");
  run<int, 0>("ds_add_i32");
  run<unsigned long long, 0>("ds_add_u64");
  run<double, 0>("ds_add_f64");
  run<float, 1>("plain rmw f32");
  return 0;
}
