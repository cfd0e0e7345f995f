// ds_read_b128 rate in msda_fwd_win's access pattern on gfx950: a quad of lanes reads 64 contiguous bytes of a 128-byte
// window slot; the four quads of a 16-lane service group take four different (16-byte half, slot parity) orders so that
// every instruction covers the 64 banks once.  Compared with the same reads WITHOUT the class rotation (bank conflicts)
// and with a plain linear sweep.  Settles DESIGN.md's "128 vs 256 B/clk/CU".
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/lds_read_bench.bin tools/micro/lds_read_bench.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((__vector_size__(16)));
typedef const f32x4 __attribute__((address_space(3)))* lds4;
constexpr int kSlots = 592, ITER = 64;

template <int MODE>   // 0: window pattern with class rotation, 1: without rotation, 2: linear sweep (lane * 16)
__global__ void __launch_bounds__(1024) reads(unsigned long long* out, float* sink, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int o = threadIdx.x * 16; o < kSlots * 128; o += blockDim.x * 16) *reinterpret_cast<f32x4*>(smem + o) = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63, k = lane & 3, pq = lane >> 2, wv = threadIdx.x >> 6;
  const int cls_a = (lane >> 3) & 1, cls_e = (lane >> 4) & 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned rnd = seed + 977u * (unsigned)(wv * 16 + pq);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    rnd = rnd * 1664525u + 1013904223u;
    unsigned slot = (rnd >> 8) % (kSlots - 24);        // top-left pixel of a "sample"; +1 and +22 stay inside
    unsigned aF, aS;
    if (MODE == 2) {
      aF = base + (unsigned)(((it * 64 + lane) * 16) % (kSlots * 128 - 4096));
      aS = aF + 1024;
    } else {
      const unsigned sw = MODE == 0 ? ((slot ^ (unsigned)cls_e) & 1u) : 0u;
      const unsigned c0 = 16u * k + (MODE == 0 ? 64u * cls_a : 0u);
      aF = base + slot * 128u + (sw << 7) + c0;
      aS = base + slot * 128u + 128u - (sw << 7) + c0;
    }
    lds4 pF = reinterpret_cast<lds4>((uintptr_t)aF), pF2 = reinterpret_cast<lds4>((uintptr_t)(aF ^ 64u));
    lds4 pS = reinterpret_cast<lds4>((uintptr_t)aS), pS2 = reinterpret_cast<lds4>((uintptr_t)(aS ^ 64u));
    const f32x4 a = pF[0], b = pF2[0], c = pS[0], d = pS2[0], e = pF[22 * 8], f = pF2[22 * 8], g = pS[22 * 8], h = pS2[22 * 8];
    acc += a + b + c + d + e + f + g + h;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc[0] == 12345.f) sink[0] = acc[1];
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + wv] = t1 - t0;
}

template <int MODE>
int run(const char* name, int cus, int threads, int wgs_per_cu, unsigned long long* dbuf, float* sink) {
  const int lds = (kSlots + 24) * 128;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(reads<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = cus * wgs_per_cu;
  hipLaunchKernelGGL(reads<MODE>, dim3(grid), dim3(threads), lds, 0, dbuf, sink, 1u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(reads<MODE>, dim3(grid), dim3(threads), lds, 0, dbuf, sink, 7u + i);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int waves = grid * threads / 64;
  std::vector<unsigned long long> h(waves);
  CHECK(hipMemcpy(h.data(), dbuf, waves * 8, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double bytes_cu = (double)wgs_per_cu * threads * ITER * 8 * 16;   // per CU and launch
  printf("%-34s %4d thr x %d WG/CU: median wave %6llu ticks for %d x 8 ds_read_b128 = %.1f B/tick/CU (whole launch incl. fill: %.1f us; %.0f KB read per CU)\n",
         name, threads, wgs_per_cu, h[waves / 2], ITER, bytes_cu / (double)h[waves / 2], ms * 1e3 / reps, bytes_cu / 1024);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  unsigned long long* dbuf; float* sink;
  CHECK(hipMalloc(&dbuf, 256 * 64 * 8)); CHECK(hipMalloc(&sink, 64));
  printf("ticks = s_memtime units (shader clocks per MI355X_MICROARCH.md; tools/micro/valu_rate prints ticks per us on this box)\n");
  for (int wg : {1, 2}) {
    run<0>("window pattern, class rotation", cus, 512, wg, dbuf, sink);
    run<1>("window pattern, NO rotation", cus, 512, wg, dbuf, sink);
    run<2>("linear sweep", cus, 512, wg, dbuf, sink);
  }
  run<0>("window pattern, class rotation", cus, 704, 2, dbuf, sink);
  return 0;
}
