// How long does a CU slot stay empty between the end of one workgroup and the start of its successor?
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/dispatch_gap.bin tools/micro/dispatch_gap.cpp && tools/micro/dispatch_gap.bin
//
// Every workgroup spins for a fixed time (s_memrealtime, 100 MHz) and records, per wave, entry time, exit time, HW_ID and
// XCC_ID.  The grid holds several rounds of workgroups per CU; per CU the i-th start beyond the resident set is paired
// with the (i - R)-th end: the difference is the dispatch gap.  Swept over the dynamic LDS size (0 ... 79 KB: one-shot
// window kernels allocate 79 KB, two workgroups per CU), the workgroup size, and with / without a trailing global store.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

struct Rec { uint64_t t0, t1; uint32_t hwid, xcc; };

__global__ void spin(Rec* rec, int ticks, int do_store, float* sink, int touch) {
  extern __shared__ char lds[];
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  if (touch) {   // the allocation is what matters to the dispatcher; one write keeps the compiler from dropping it
    lds[threadIdx.x * 4] = (char)threadIdx.x;
    __syncthreads();
  }
  while ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) < (int64_t)ticks) __builtin_amdgcn_s_sleep(2);
  if (do_store) {
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (float)lds[(threadIdx.x * 4) & 1023];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    Rec r;
    r.t0 = t0; r.t1 = t1;
    r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID, all 32 bits
    r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
    rec[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
  }
}

static double pct(std::vector<double>& v, double p) {
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  return v[std::min(v.size() - 1, (size_t)(p * (v.size() - 1) + 0.5))];
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("%s: %d CUs; realtime ticks of 10 ns\n", prop.gcnArchName, cus);
  const int spin_ticks = 1000;   // 10 us
  const int lds_sizes[] = {0, 32768, 65536, 66560, 80896};
  const int thread_counts[] = {256, 512};
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int threads : thread_counts) {
    for (int ldsb : lds_sizes) {
      for (int do_store = 0; do_store < 2; ++do_store) {
        const int waves = threads / 64;
        int by_lds = ldsb ? (160 * 1024) / ldsb : 64, by_waves = 32 / waves;
        const int resident = std::max(1, std::min(by_lds, by_waves));
        const int rounds = 6;
        const int grid = cus * resident * rounds;
        Rec* rec;
        float* sink;
        CHECK(hipMalloc(&rec, sizeof(Rec) * (size_t)grid * waves));
        CHECK(hipMalloc(&sink, sizeof(float) * (size_t)grid * threads));
        std::vector<Rec> h((size_t)grid * waves);
        double span_us = 0;
        std::vector<double> gaps, wg_life;
        double mean_resident = 0;
        for (int rep = 0; rep < 3; ++rep) {   // the last repetition is analysed
          hipLaunchKernelGGL(spin, dim3(grid), dim3(threads), ldsb, 0, rec, spin_ticks, do_store, sink, ldsb ? 1 : 0);
          CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h.data(), rec, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost));
        // per workgroup: first wave in, last wave out, and the CU it ran on
        struct Wg { uint64_t s, e; };
        std::map<uint32_t, std::vector<Wg>> per_cu;
        uint64_t tmin = ~0ull, tmax = 0;
        double life_sum = 0;
        for (int g = 0; g < grid; ++g) {
          Wg w{~0ull, 0};
          for (int v = 0; v < waves; ++v) {
            const Rec& r = h[(size_t)g * waves + v];
            w.s = std::min(w.s, r.t0);
            w.e = std::max(w.e, r.t1);
          }
          const Rec& r0 = h[(size_t)g * waves];
          const uint32_t cu = (r0.hwid >> 8) & 0xf, sh = (r0.hwid >> 12) & 1, se = (r0.hwid >> 13) & 7, xcc = r0.xcc & 0xf;
          per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(w);
          tmin = std::min(tmin, w.s);
          tmax = std::max(tmax, w.e);
          life_sum += (double)(w.e - w.s);
          wg_life.push_back((w.e - w.s) * 0.01);
        }
        span_us = (tmax - tmin) * 0.01;
        mean_resident = life_sum / (double)(tmax - tmin);
        int max_res = 0;
        for (auto& kv : per_cu) {
          auto& v = kv.second;
          std::sort(v.begin(), v.end(), [](const Wg& a, const Wg& b) { return a.s < b.s; });
          std::vector<uint64_t> ends;
          for (auto& w : v) ends.push_back(w.e);
          std::sort(ends.begin(), ends.end());
          int R = 0;
          while (R < (int)v.size() && v[R].s < ends[0]) ++R;
          max_res = std::max(max_res, R);
          for (size_t i = R; i < v.size(); ++i) gaps.push_back(((double)v[i].s - (double)ends[i - R]) * 0.01);
        }
        std::vector<double> g2 = gaps, l2 = wg_life;
        std::printf("threads %3d  lds %6d B  store %d: CUs seen %3zu, resident/CU %d (expected %d), grid %5d, span %7.2f us (ideal %6.2f), "
                    "mean resident %6.1f of %d; workgroup life median %5.2f us; dispatch gap median %5.2f  p10 %5.2f  p90 %5.2f us (%zu gaps)\n",
                    threads, ldsb, do_store, per_cu.size(), max_res, resident, grid, span_us, rounds * 10.0, mean_resident,
                    cus * resident, pct(l2, 0.5), pct(g2, 0.5), pct(g2, 0.1), pct(g2, 0.9), gaps.size());
        CHECK(hipFree(rec));
        CHECK(hipFree(sink));
      }
    }
  }
  return 0;
}
