// Micro-benchmark: is one of the eight 128-byte residues of a 1 KB-strided gather slower than the others?
// (value[N, S, 8 heads, 32] f32: the lines of head m are the lines with address bits 9:7 == m.)
//   hipcc -O3 --offload-arch=gfx950 -o stride_residue_bench.bin stride_residue_bench.cpp
// Modes: "all r": every workgroup gathers lines of residue r;  "xcd": workgroup b gathers residue (b + rot) % 8
// (workgroups go to the 8 XCDs round-robin, so each XCD owns one residue -- the msda_fwd_lg3 mapping).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((__vector_size__(16)));

__global__ void __launch_bounds__(256) gather(const char* __restrict__ buf, int npix, int mode, int rot, int iters,
                                              int local, float* __restrict__ sink) {
  const int lane8 = threadIdx.x & 7, grp = threadIdx.x >> 3;   // 32 groups of 8 lanes, one 128-B line per group
  int res = mode == 0 ? rot : ((blockIdx.x + rot) & 7);   // modes 1..3: one residue / head per XCD
  unsigned p = (blockIdx.x >> 3) * 2654435761u + grp * 40503u;
  const int base_pix = (int)(((long long)(blockIdx.x >> 3) * npix) / (gridDim.x >> 3));   // local mode: own band
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
    if (mode == 2) {   // Latin square in time: the XCD's residue changes from segment to segment
      const int sg = i * 8 / iters;
      res = (blockIdx.x + rot + (((sg & 1) << 2) | (sg & 2) | ((sg >> 2) & 1))) & 7;
    }
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      p = p * 1664525u + 1013904223u;
      int pix = (local & 1) ? (base_pix + (int)((p >> 8) % 4096u)) % npix : (int)((p >> 8) % (unsigned)npix);
      const size_t off = mode == 3 ? ((size_t)res * npix + pix) * 128     // head-major layout: dense slice per residue
                                   : (size_t)pix * 1024 + res * 128;
      const f32x4* ptr = reinterpret_cast<const f32x4*>(buf + off + lane8 * 16);
      v[u] = (local & 2) ? __builtin_nontemporal_load(ptr) : *ptr;   // bit 1 of `local`: non-temporal (L1-bypassing) loads
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

int main(int argc, char** argv) {
  const int npix = 2 * 22223;
  const size_t bytes = (size_t)npix * 1024 + 4096;
  char* buf;
  float* sink;
  hipMalloc(&buf, bytes);
  hipMalloc(&sink, 64);
  hipMemset(buf, 0, bytes);
  const int shift = argc > 1 ? atoi(argv[1]) : 0;   // shift the base by this many bytes
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 8 * 256, iters = 32;   // 2048 workgroups x 32 groups x 256 lines
  auto run = [&](int mode, int rot, int local) {
    for (int w = 0; w < 2; ++w) gather<<<blocks, 256>>>(buf + shift, npix, mode, rot, iters, local, sink);
    hipEventRecord(a);
    for (int w = 0; w < 5; ++w) gather<<<blocks, 256>>>(buf + shift, npix, mode, rot, iters, local, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5 * 1e3;
  };
  const double lines = (double)blocks * 32 * iters * 8;
  for (int local = 0; local < 4; ++local) {
    printf("%s pixels%s, base shift %d B: %0.f lines of 128 B per launch\n", (local & 1) ? "band-local" : "random",
           (local & 2) ? ", NON-TEMPORAL loads" : "", shift, lines);
    for (int r = 0; r < 8; ++r) {
      const float us = run(0, r, local);
      printf("  all workgroups residue %d: %8.1f us  %6.2f TB/s\n", r, us, lines * 128 / us * 1e-6);
    }
    for (int rot = 0; rot < 8; rot += 3) {
      const float us = run(1, rot, local);
      printf("  one residue per XCD (rot %d): %8.1f us  %6.2f TB/s\n", rot, us, lines * 128 / us * 1e-6);
    }
    for (int rot = 0; rot < 8; rot += 3) {
      const float us = run(3, rot, local);
      printf("  head-major layout (dense 5.7 MB slice per XCD) (rot %d): %8.1f us  %6.2f TB/s\n", rot, us, lines * 128 / us * 1e-6);
    }
    for (int rot = 0; rot < 8; rot += 3) {
      const float us = run(2, rot, local);
      printf("  residue rotating over 8 time segments per XCD (rot %d): %8.1f us  %6.2f TB/s\n", rot, us, lines * 128 / us * 1e-6);
    }
  }
  return 0;
}
