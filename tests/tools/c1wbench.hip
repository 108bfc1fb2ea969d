// Stand-alone timing of conv1_wgrad_shift_kernel on synthetic buffers (no engine around it): back-to-back launches on
// the same 4096-sample window (MALL-warm) and on rotating windows (cold), plus larger batches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tests/tools/c1wbench.hip -o build_tools/c1wbench
#include "../../ale-libtorch-ppo_amd/csrc/common.hpp"
#include "../../ale-libtorch-ppo_amd/csrc/conv1_wgrad.hpp"
#include <cstdio>
#include <cstdlib>
using namespace aleppo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void fill_kernel(uint32_t *p, size_t n, uint32_t seed, uint32_t mask, uint32_t orv) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (x & mask) | orv;
  }
}
__global__ void icache_inv_kernel() { asm volatile("s_icache_inv\n s_nop 7\n s_nop 7" ::: "memory"); }
int main(int argc, char **argv) {
  const bool rnd = argc > 1;
  const long NS = 65536;
  uint8_t *obs; bf16 *dy; float *sw, *sb;
  CK(hipMalloc(&obs, NS * 28224 + 65536));
  CK(hipMalloc(&dy, NS * 25600 + 65536));
  CK(hipMalloc(&sw, 256 * 32 * 256 * 4));
  CK(hipMalloc(&sb, 256 * 32 * 4));
  CK(hipMemset(obs, 3, NS * 28224));
  CK(hipMemset(dy, 0, NS * 25600));
  if (rnd) { // random frame bytes; dY = bf16 values of magnitude ~1e-3 .. 1e-2 with random signs / mantissas
    fill_kernel<<<2048, 256>>>((uint32_t *)obs, NS * 28224 / 4, 1u, 0xFFFFFFFFu, 0u);
    fill_kernel<<<2048, 256>>>((uint32_t *)dy, NS * 25600 / 4, 2u, 0x80FF80FFu, 0x3B003B00u);
    CK(hipDeviceSynchronize());
    std::printf("random data\n");
  }
  CK(hipFuncSetAttribute((const void *)conv1_wgrad_shift_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c1w::SMEM));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, long ns, int reps, bool rotate) {
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) {
        const int n0 = rotate ? (int)((r * ns) % (NS - ns + 1)) : 0;
        WgradParams P{obs, dy + (long)n0 * 12800, sw, sb, ns, SampleMap{1, 0, 7056, 0, n0}, 1.0f / 255.0f};
        hipLaunchKernelGGL(conv1_wgrad_shift_kernel, dim3(256), dim3(c1w::NTHREADS), c1w::SMEM, 0, P);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    std::printf("%-28s ns %6ld: %8.1f us  %5.2f TB/s\n", name, ns, us, (double)ns * (28224 + 25600) / us * 1e-6);
  };
  // as inside the update: a kernel that has just WRITTEN the dY window runs before every launch; only the wgrad is timed
  {
    float tot = 0;
    const int reps = 16;
    for (int r = 0; r < reps + 2; ++r) {
      fill_kernel<<<2048, 256>>>((uint32_t *)dy, 4096 * 25600 / 4, 2u + r, 0x80FF80FFu, 0x3B003B00u);
      CK(hipEventRecord(e0));
      WgradParams P{obs, dy, sw, sb, 4096, SampleMap{1, 0, 7056, 0, 0}, 1.0f / 255.0f};
      hipLaunchKernelGGL(conv1_wgrad_shift_kernel, dim3(256), dim3(c1w::NTHREADS), c1w::SMEM, 0, P);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) tot += ms;
    }
    std::printf("after a producer of dY, events around the launch: %8.1f us\n", tot * 1e3 / reps);
    tot = 0;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipEventRecord(e0));
      WgradParams P{obs, dy, sw, sb, 4096, SampleMap{1, 0, 7056, 0, 0}, 1.0f / 255.0f};
      hipLaunchKernelGGL(conv1_wgrad_shift_kernel, dim3(256), dim3(c1w::NTHREADS), c1w::SMEM, 0, P);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) tot += ms;
    }
    std::printf("no producer, events around the launch (sync each): %8.1f us\n", tot * 1e3 / reps);
  }
  for (int mode = 0; mode < 3; ++mode) { // 0: as is; 1: L2 / MALL flushed by a 1.2 GB write before every launch; 2: + I-cache invalidated
    float tot = 0;
    const int reps = 12;
    for (int r = 0; r < reps + 2; ++r) {
      if (mode >= 1)
        fill_kernel<<<2048, 256>>>((uint32_t *)dy + (size_t)8192 * 6400, (size_t)300 << 20, 2u + r, 0x80FF80FFu, 0x3B003B00u);
      if (mode >= 2)
        icache_inv_kernel<<<2048, 64>>>();
      CK(hipEventRecord(e0));
      WgradParams P{obs, dy, sw, sb, 4096, SampleMap{1, 0, 7056, 0, 0}, 1.0f / 255.0f};
      hipLaunchKernelGGL(conv1_wgrad_shift_kernel, dim3(256), dim3(c1w::NTHREADS), c1w::SMEM, 0, P);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) tot += ms;
    }
    std::printf("mode %d (0 warm, 1 caches flushed by a 1.2 GB write, 2 + s_icache_inv): %8.1f us\n", mode, tot * 1e3 / reps);
  }
  run("same window (MALL-warm)", 4096, 16, false);
  run("rotating windows (cold)", 4096, 16, true);
  run("rotating windows (cold)", 8192, 8, true);
  run("rotating windows (cold)", 16384, 4, true);
  run("whole buffer", 65536, 2, false);

  return 0;
}
