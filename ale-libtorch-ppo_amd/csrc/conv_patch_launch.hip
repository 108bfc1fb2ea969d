// conv_patch_launch.hip - launch geometry of the sample-stationary bf16 conv kernels (conv_patch.hpp).
// Persistent grids: at most one workgroup per CU (the LDS patch buffers are 41..155 KB), 512 threads.
#include "common.hpp"
#include "conv_patch.hpp"
#include "conv1_wgrad.hpp"
#include "conv3_tile.hpp"
#include "conv_fwd_fused.hpp"
#include "conv_bwd_fused.hpp"
#include <cstdio>
#include <cstdlib>

namespace aleppo {

static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      n = p.multiProcessorCount;
    if (n <= 0)
      n = 256;
  }
  return n;
}

// developer switches that select timing-only variants of the fused kernels (a phase left out: WRONG results): say so, loudly
static int ablation_switch(const char *name) {
  const char *e = std::getenv(name);
  const int v = e ? std::atoi(e) : 0;
  if (v)
    std::fprintf(stderr, "aleppo: %s=%d selects a TIMING-ONLY kernel variant: the update's results are WRONG\n", name, v);
  return v;
}

template <class K> static void allow_smem(K kernel, size_t bytes) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes);
}

// scratch that forward lanes without an output pixel store to (keeps the store count per group static)
static bf16 *fwd_dummy() {
  static bf16 *p[16] = {nullptr};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  if (!p[dev] && hipMalloc(reinterpret_cast<void **>(&p[dev]), 16 * 1024) != hipSuccess)
    p[dev] = nullptr; // (a null scratch would fault: allocation failure is reported by the next HIP call)
  return p[dev];
}

// NW waves per workgroup, WPC persistent workgroups per CU
template <class L, int NW, int WPC> static void launch_patch(hipStream_t s, const PatchParams &P) {
  static bool once = false;
  constexpr size_t sm = conv_patch_smem<L>();
  static_assert(sm * WPC <= 160 * 1024, "LDS budget per CU");
  if (!once) {
    allow_smem(conv_patch_kernel<L, NW>, sm);
    once = true;
  }
  const long ngroups = (P.ns * L::GPS + L::SB - 1) / L::SB;
  const int grid = (int)std::min<long>(ngroups, (long)num_cus() * WPC);
  hipLaunchKernelGGL((conv_patch_kernel<L, NW>), dim3(grid), dim3(64 * NW), sm, s, P);
}

void patch_conv1_fwd(hipStream_t s, const uint32_t *obs, SampleMap map, const void *W1, const float *b1, void *a1,
                     long ns) {
  PatchParams P{obs, static_cast<const bf16 *>(W1), b1, nullptr, static_cast<bf16 *>(a1), ns, map, 1.0f / 255.0f, fwd_dummy()};
  launch_patch<LConv1Fwd, 8, 2>(s, P);
}
void patch_conv2_fwd(hipStream_t s, const void *a1, const void *W2, const float *b2, void *a2, long ns) {
  PatchParams P{a1, static_cast<const bf16 *>(W2), b2, nullptr, static_cast<bf16 *>(a2), ns, SampleMap{1, 0, 0, 0, 0},
                1.0f, fwd_dummy()};
  if (ns <= 256)
    launch_patch<LConv2FwdSmall, 8, 1>(s, P);
  else
    launch_patch<LConv2FwdW4, 4, 2>(s, P);
}
void patch_conv3_fwd(hipStream_t s, const void *a2, const void *W3, const float *b3, void *a3, long ns) {
  PatchParams P{a2, static_cast<const bf16 *>(W3), b3, nullptr, static_cast<bf16 *>(a3), ns, SampleMap{1, 0, 0, 0, 0},
                1.0f, fwd_dummy()};
  if (ns <= 256)
    launch_patch<LConv3FwdSmall, 8, 1>(s, P);
  else
    launch_patch<LConv3Fwd1W4, 4, 2>(s, P);
}
template <int ABL> static void launch_fwd_fused(hipStream_t s, const FwdFusedParams &P) {
  static bool once = false;
  if (!once) {
    allow_smem(fwd_fused_kernel<ABL>, FWD_FUSED_SMEM);
    once = true;
  }
  hipLaunchKernelGGL(fwd_fused_kernel<ABL>, dim3((unsigned)std::min<long>(P.ns, num_cus())), dim3(FF_NT), FWD_FUSED_SMEM, s, P);
}
void patch_fwd_fused(hipStream_t s, const uint32_t *obs, SampleMap map, const void *W1, const float *b1, const void *W2,
                     const float *b2, const void *W3, const float *b3, void *a1, void *a2, void *a3, long ns) {
  FwdFusedParams P{obs, map, static_cast<const bf16 *>(W1), static_cast<const bf16 *>(W2), static_cast<const bf16 *>(W3),
                   b1,  b2,  b3, static_cast<bf16 *>(a1), static_cast<bf16 *>(a2), static_cast<bf16 *>(a3), ns};
  // ALEPPO_FF_ABLATE: timing-only builds of the kernel with one part left out (wrong results)
  static const int abl = ablation_switch("ALEPPO_FF_ABLATE");
  switch (abl) {
  case 1: return launch_fwd_fused<1>(s, P);
  case 2: return launch_fwd_fused<2>(s, P);
  case 4: return launch_fwd_fused<4>(s, P);
  case 8: return launch_fwd_fused<8>(s, P);
  case 16: return launch_fwd_fused<16>(s, P);

  default: return launch_fwd_fused<0>(s, P);
  }
}
template <int ABL> static void launch_conv_bwd(hipStream_t s, const ConvBwdParams &P, int grid) {
  static bool once = false;
  if (!once) {
    allow_smem(conv_bwd_fused_kernel<ABL>, cb::SMEM);
    once = true;
  }
  hipLaunchKernelGGL(conv_bwd_fused_kernel<ABL>, dim3(grid), dim3(cb::NT), cb::SMEM, s, P);
}
int patch_conv_bwd_fused(hipStream_t s, const void *dz2, const void *a1, const uint32_t *obs, SampleMap map, const void *W2d,
                         float *sw2, float *sb2, float *sw1, float *sb1, long ns) {
  ConvBwdParams P{static_cast<const bf16 *>(dz2), static_cast<const bf16 *>(a1), static_cast<const bf16 *>(W2d), obs, map,
                  sw2, sb2, sw1, sb1, ns, 1.0f / 255.0f};
  const int grid = (int)std::min<long>(ns, std::min(num_cus(), std::min(MAXS_C1, MAXS_C2)));
  // ALEPPO_CB_ABLATE: timing-only builds of the kernel with one part left out (wrong results)
  static const int abl = ablation_switch("ALEPPO_CB_ABLATE");
  switch (abl) {
  case 1: launch_conv_bwd<1>(s, P, grid); break;
  case 2: launch_conv_bwd<2>(s, P, grid); break;
  case 4: launch_conv_bwd<4>(s, P, grid); break;
  case 8: launch_conv_bwd<8>(s, P, grid); break;
  case 16: launch_conv_bwd<16>(s, P, grid); break;
  case 31: launch_conv_bwd<31>(s, P, grid); break;
  default: launch_conv_bwd<0>(s, P, grid); break;
  }
  return grid;
}
void patch_conv3_dgrad(hipStream_t s, const void *dz3, const void *W3d, const void *a2, void *dz2, long ns) {
  PatchParams P{dz3, static_cast<const bf16 *>(W3d), nullptr, static_cast<const bf16 *>(a2), static_cast<bf16 *>(dz2),
                ns,  SampleMap{1, 0, 0, 0, 0},       1.0f, nullptr};
  static const int tile = [] {
    const char *e = std::getenv("ALEPPO_C3D_TILE"); // =0: the sample-stationary kernel (A/B testing: 46 vs 40 us)
    return e ? std::atoi(e) : 1;
  }();
  if (tile) {
    static bool once = false;
    if (!once) {
      allow_smem(conv3_dgrad_tile_kernel, C3T_SMEM);
      once = true;
    }
    C3TileParams T{static_cast<const bf16 *>(dz3), static_cast<const bf16 *>(W3d), static_cast<const bf16 *>(a2),
                   static_cast<bf16 *>(dz2), fwd_dummy(), ns};
    const long ngroups = (ns + C3T_SB - 1) / C3T_SB;
    hipLaunchKernelGGL(conv3_dgrad_tile_kernel, dim3((unsigned)std::min<long>(ngroups, num_cus())), dim3(512), C3T_SMEM, s,
                       T);
    return;
  }
  launch_patch<LConv3Dgrad, 8, 1>(s, P); // the 4-wave / 2-per-CU variant measured slower here (56 vs 48 us)
}
void patch_conv2_dgrad(hipStream_t s, const void *dz2, const void *W2d, const void *a1, void *dz1, long ns) {
  PatchParams P{dz2, static_cast<const bf16 *>(W2d), nullptr, static_cast<const bf16 *>(a1), static_cast<bf16 *>(dz1),
                ns,  SampleMap{1, 0, 0, 0, 0},       1.0f, nullptr};
  launch_patch<LConv2DgradW4, 4, 2>(s, P); // 2 per CU: the preloaded gates take the kernel to 196 VGPRs
}

void patch_act_convs(hipStream_t s, uint32_t *obs, SampleMap map, const void *W1, const float *b1, const void *W2,
                     const float *b2, const void *W3, const float *b3, void *a3, long ns, int ingest_mode,
                     const uint8_t *frames, const uint8_t *lut, const StartBits *sbits, long src_delta,
                     const uint8_t *start_bytes) {
  static bool once = false;
  if (!once) {
    allow_smem(act_conv_kernel<0>, ACT_SMEM);
    allow_smem(act_conv_kernel<1>, ACT_SMEM);
    allow_smem(act_conv_kernel<2>, ACT_SMEM);
    once = true;
  }
  ActConvParams P{obs, map, static_cast<const bf16 *>(W1), static_cast<const bf16 *>(W2), static_cast<const bf16 *>(W3),
                  b1,  b2,  b3, static_cast<bf16 *>(a3), ns};
  ActIngestParams G{};
  G.frames = frames;
  G.lut = lut;
  G.obs_rw = obs;
  G.src_delta = src_delta;
  if (sbits)
    G.sbits = *sbits;
  G.start_bytes = start_bytes;
  const dim3 g((unsigned)std::min<long>(ns, num_cus())), b(512);
  if (ingest_mode == 1)
    hipLaunchKernelGGL(act_conv_kernel<1>, g, b, ACT_SMEM, s, P, G);
  else if (ingest_mode == 2)
    hipLaunchKernelGGL(act_conv_kernel<2>, g, b, ACT_SMEM, s, P, G);
  else
    hipLaunchKernelGGL(act_conv_kernel<0>, g, b, ACT_SMEM, s, P, G);
}

template <class L> static int launch_wgrad(hipStream_t s, const WgradParams &P) {
  static bool once = false;
  constexpr size_t sm = conv_wgrad_patch_smem<L>();
  if (!once) {
    allow_smem(conv_wgrad_patch_kernel<L>, sm);
    once = true;
  }
  const long ngroups = (P.ns * L::GPS + L::SB - 1) / L::SB;
  // Half the CUs: every workgroup writes one fp32 slab of the whole dW (147 KB for conv3), and these kernels run on the
  // weight-gradient stream in the shadow of the dgrad chain, so fewer, longer workgroups cost nothing there and halve what
  // the slab reduce reads beside conv1 wgrad (measured per 4096-sample minibatch, same box: 256 workgroups 455 us,
  // 192: 451, 160: 450, 128: 444, 96: 455; ALEPPO_WG_GRID overrides)
  static const int cap = [] {
    const char *e = std::getenv("ALEPPO_WG_GRID");
    return e ? std::atoi(e) : num_cus() / 2;
  }();
  const int grid = (int)std::min<long>(ngroups, std::min(std::min(num_cus(), MAXS_C1), cap));
  hipLaunchKernelGGL((conv_wgrad_patch_kernel<L>), dim3(grid), dim3(512), sm, s, P);
  return grid;
}
int patch_conv1_wgrad(hipStream_t s, const void *dz1, const uint32_t *obs, SampleMap map, float *sw, float *sb,
                      long ns) {
  WgradParams P{obs, static_cast<const bf16 *>(dz1), sw, sb, ns, map, 1.0f / 255.0f};
  static bool once = false;
  if (!once) {
    allow_smem(conv1_wgrad_shift_kernel, c1w::SMEM);
    once = true;
  }
  const int grid = (int)std::min<long>(2 * ns, std::min(num_cus(), MAXS_C1));
  hipLaunchKernelGGL(conv1_wgrad_shift_kernel, dim3(grid), dim3(c1w::NTHREADS), c1w::SMEM, s, P);
  return grid;
}
int patch_conv2_wgrad(hipStream_t s, const void *dz2, const void *a1, float *sw, float *sb, long ns) {
  WgradParams P{a1, static_cast<const bf16 *>(dz2), sw, sb, ns, SampleMap{1, 0, 0, 0, 0}, 1.0f};
  return launch_wgrad<LConv2Wgrad>(s, P);
}
int patch_conv3_wgrad(hipStream_t s, const void *dz3, const void *a2, float *sw, float *sb, long ns) {
  WgradParams P{a2, static_cast<const bf16 *>(dz3), sw, sb, ns, SampleMap{1, 0, 0, 0, 0}, 1.0f};
  return launch_wgrad<LConv3Wgrad>(s, P);
}

} // namespace aleppo
