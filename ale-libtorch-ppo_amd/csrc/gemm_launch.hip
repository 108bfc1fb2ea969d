// gemm_launch.hip - instantiations + launch geometry of the conv / linear stack kernels (gemm.hpp).
// Layer shapes follow NetworkImpl (reference src/bin/train.cc:232-244); activations are NHWC,
// weights [oc][(kh,kw,c)].  No vendor BLAS / MIOpen on this path.
#include "common.hpp"
#include "gemm.hpp"
#include "gemm_pipe.hpp"
#include <algorithm>
#include <cstdlib>

namespace aleppo {

// tile-shape A/B switches: every call site keeps its value in a function-local static (read once per process); the
// switches the tests flip at run time (fc_pipe, patch_conv) are per-context options (tuning())
static int tune(const char *name, int dflt) {
  const char *e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
static inline dim3 grid2(long M, int BM, long N, int BN, int Z = 1) {
  return dim3((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN), (unsigned)Z);
}

// ------------------------------------------------------------------ forward
template <class T>
static void conv1_fwd_t(hipStream_t s, const uint32_t *obs, SampleMap map, const void *W1, const float *b1, void *a1,
                        long ns) {
  // A: packed u8 stack [84][84][4] per sample, 8x8 s4 window -> k = (kh, kw*4+c), 32 contiguous bytes per kh
  using AL = ConvGatherLoader<T, uint8_t, 400, 20, 4, 84, 4, 8>;
  using BL = DenseLoader<T>;
  using EP = EpiBiasAct<T, true>;
  const long M = ns * 400;
  typename AL::P ap{reinterpret_cast<const uint8_t *>(obs), map.TP, map.s1 * 4, map.s0 * 4, map.base * 4, map.n0};
  typename BL::P bp{static_cast<const T *>(W1), 256, 0};
  typename EP::P ep{static_cast<T *>(a1), b1, 32, 1.0f / 255.0f}; // x/255 folded into the epilogue (train.cc:258-259)
  hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 32, 4, 1>), grid2(M, 128, 32, 32), dim3(256), 0, s, ap, bp,
                     ep, (int)M, 32, 256);
}
template <class T>
static void conv2_fwd_t(hipStream_t s, const void *a1, const void *W2, const float *b2, void *a2, long ns) {
  using AL = ConvGatherLoader<T, T, 81, 9, 2, 20, 32, 4>;
  using BL = DenseLoader<T>;
  using EP = EpiBiasAct<T, true>;
  const long M = ns * 81;
  typename AL::P ap{static_cast<const T *>(a1), 1, 400 * 32, 0, 0, 0};
  typename BL::P bp{static_cast<const T *>(W2), 512, 0};
  typename EP::P ep{static_cast<T *>(a2), b2, 64, 1.0f};
  hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 64, 2, 2>), grid2(M, 128, 64, 64), dim3(256), 0, s, ap, bp,
                     ep, (int)M, 64, 512);
}
template <class T>
static void conv3_fwd_t(hipStream_t s, const void *a2, const void *W3, const float *b3, void *a3, long ns) {
  using AL = ConvGatherLoader<T, T, 49, 7, 1, 9, 64, 3>;
  using BL = DenseLoader<T>;
  using EP = EpiBiasAct<T, true>;
  const long M = ns * 49;
  typename AL::P ap{static_cast<const T *>(a2), 1, 81 * 64, 0, 0, 0};
  typename BL::P bp{static_cast<const T *>(W3), 576, 0};
  typename EP::P ep{static_cast<T *>(a3), b3, 64, 1.0f};
  hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 64, 2, 2>), grid2(M, 128, 64, 64), dim3(256), 0, s, ap, bp,
                     ep, (int)M, 64, 576);
}

// ------------------------------------------------------------------ pipelined bf16 fc GEMMs (gemm_pipe.hpp)
static int pipe_cus() {
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
template <int MODE, int NST> static void launch_pipe(hipStream_t s, const PipeParams &P) {
  static bool once = false;
  constexpr size_t sm = gemm_pipe_smem<MODE, NST>();
  static_assert(sm <= 160 * 1024, "LDS budget");
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pipe_kernel<MODE, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    once = true;
  }
  // Persistent workgroups, at most one per CU: every workgroup gets the SAME number of jobs (800 dgrad jobs run as
  // 200 workgroups x 4 rather than 256 workgroups of which 32 do a 4th round), which takes the same time and leaves the
  // other CUs to the weight-gradient kernel co-scheduled on the second stream.  A multiple of 8 so that the
  // XCD-grouped job maps apply; jobless workgroups exit at once.
  const int cus = pipe_cus() / 8 * 8, rounds = (P.njobs + cus - 1) / cus;
  const int grid = P.njobs >= cus / 4 ? std::min(cus, ((P.njobs + rounds - 1) / rounds + 7) / 8 * 8) : std::min(P.njobs, cus);
  hipLaunchKernelGGL((gemm_pipe_kernel<MODE, NST>), dim3(grid), dim3(PIPE_THREADS), sm, s, P);
}
static bool fc_dma() {
  static const bool v = tune("ALEPPO_FC_DMA", 1) != 0;
  return v;
}
static bool tn_xcd() {
  static const bool v = tune("ALEPPO_TN_XCD", 1) != 0;
  return v;
}
static bool use_pipe() { return tuning().fc_pipe; }
// forward: h partial slabs [parts][ns][H]; parts chosen so that the jobs fill the CUs with the fewest rounds
static int fc_fwd_pipe(hipStream_t s, const void *a3, const void *Wfc, const float *bfc, float *h, long ns, int H,
                       int max_parts) {
  const int tm = (int)((ns + 127) / 128), tn = (H + 127) / 128, tiles = tm * tn, cus = pipe_cus();
  int best = 1;
  double best_t = 1e30;
  for (int sp = 1; sp <= max_parts; ++sp) {
    const double t = (double)((tiles * sp + cus - 1) / cus) / sp + 0.02 * sp; // rounds x K-share (+ slab traffic)
    if (t < best_t - 1e-9) {
      best_t = t;
      best = sp;
    }
  }
  PipeParams P{};
  P.A = static_cast<const bf16 *>(a3);
  P.lda = FC_IN;
  P.B = static_cast<const bf16 *>(Wfc);
  P.ldb = FC_IN;
  P.M = (int)ns;
  P.N = H;
  P.tiles_m = tm;
  P.tiles_n = tn;
  P.nstages = FC_IN / 64;
  P.splits = best;
  P.njobs = tiles * best;
  P.out_f32 = h;
  P.bias = bfc;
  launch_pipe<0, 4>(s, P); // ring depth 3 / 4 / 5 measured equal: the loop runs at the L2 -> LDS DMA rate
  return best;
}
static void fc_dgrad_pipe(hipStream_t s, const void *dh, const void *WfcT, const void *a3, void *dz3, long ns, int H) {
  PipeParams P{};
  P.A = static_cast<const bf16 *>(dh);
  P.lda = H;
  P.B = static_cast<const bf16 *>(WfcT);
  P.ldb = H;
  P.M = (int)ns;
  P.N = FC_IN;
  P.tiles_m = (int)((ns + 127) / 128);
  P.tiles_n = (FC_IN + 127) / 128;
  P.nstages = H / 64;
  P.splits = 1;
  P.njobs = P.tiles_m * P.tiles_n;
  P.out_bf16 = static_cast<bf16 *>(dz3);
  P.gate = static_cast<const bf16 *>(a3);
  launch_pipe<1, 4>(s, P);
}
template <class T>
static void fc_fwd_t(hipStream_t s, const void *a3, const void *Wfc, const float *bfc, float *h, long ns, int H) {
  using AL = DenseLoader<T>;
  using BL = DenseLoader<T>;
  using EP = EpiBiasAct<float, false>; // NO relu after the 3136->H linear
  typename AL::P ap{static_cast<const T *>(a3), FC_IN, 0};
  typename BL::P bp{static_cast<const T *>(Wfc), FC_IN, 0};
  typename EP::P ep{h, bfc, H, 1.0f};
  if (ns <= 256) // acting batch: smaller M tile so more workgroups share the 3136-deep reduction
    hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 32, 32, 2, 2>), grid2(ns, 32, H, 32), dim3(256), 0, s, ap, bp,
                       ep, (int)ns, H, FC_IN);
  else if (fc_dma() && FC_IN % Atom<T>::KT == 0) {
    static const int v = tune("ALEPPO_FC_FWD_TILE", 1);
    const T *a = static_cast<const T *>(a3), *b = static_cast<const T *>(Wfc);
    if (v == 3 && FC_IN % (2 * Atom<T>::KT) == 0)
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 64, 64, 2, 2, 16>), grid2(ns, 64, H, 64), dim3(256), 0, s, a, FC_IN,
                         b, FC_IN, ep, (int)ns, H, FC_IN);
    else if (v == 4 && FC_IN % (2 * Atom<T>::KT) == 0)
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 128, 64, 2, 2, 16>), grid2(ns, 128, H, 64), dim3(256), 0, s, a,
                         FC_IN, b, FC_IN, ep, (int)ns, H, FC_IN);
    else if (v == 1)
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 64, 64, 2, 2>), grid2(ns, 64, H, 64), dim3(256), 0, s, a, FC_IN, b,
                         FC_IN, ep, (int)ns, H, FC_IN);
    else if (v == 2)
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 128, 128, 2, 2>), grid2(ns, 128, H, 128), dim3(256), 0, s, a,
                         FC_IN, b, FC_IN, ep, (int)ns, H, FC_IN);
    else
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 128, 64, 2, 2>), grid2(ns, 128, H, 64), dim3(256), 0, s, a, FC_IN,
                         b, FC_IN, ep, (int)ns, H, FC_IN);
  } else {
    static const int v = tune("ALEPPO_FC_FWD_TILE", 1);
    if (v == 1)
      hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 64, 64, 2, 2>), grid2(ns, 64, H, 64), dim3(256), 0, s, ap, bp,
                         ep, (int)ns, H, FC_IN);
    else if (v == 2)
      hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 128, 2, 2>), grid2(ns, 128, H, 128), dim3(256), 0, s, ap,
                         bp, ep, (int)ns, H, FC_IN);
    else if (v == 3)
      hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 64, 128, 2, 2>), grid2(ns, 64, H, 128), dim3(256), 0, s, ap,
                         bp, ep, (int)ns, H, FC_IN);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 64, 2, 2>), grid2(ns, 128, H, 64), dim3(256), 0, s, ap,
                         bp, ep, (int)ns, H, FC_IN);
  }
}

// acting-size fc (ns <= 256 rows): split the 3136-deep reduction over FC_SPLITS grid.z slices so 100s of
// workgroups share it; slices land in hpart[z][ns][H] and the head kernel adds them (+ bias) in fixed order
template <class T> static void fc_fwd_splitk_t(hipStream_t s, const void *a3, const void *Wfc, float *hpart, long ns, int H) {
  using AL = DenseLoader<T>;
  using BL = DenseLoader<T>;
  using EP = EpiPartial;
  constexpr int KC = FC_IN / FC_SPLITS; // 448 = 7*64 = 14*32
  typename AL::P ap{static_cast<const T *>(a3), FC_IN, KC};
  typename BL::P bp{static_cast<const T *>(Wfc), FC_IN, KC};
  typename EP::P ep{hpart, H, ns * (long)H};
  hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 32, 32, 2, 2>), grid2(ns, 32, H, 32, FC_SPLITS), dim3(256), 0, s,
                     ap, bp, ep, (int)ns, H, KC);
}
// bf16 acting fc, latency form.  At acting size (128 rows) the GEMM is a dependency chain, not a throughput problem: the
// staged kernel above walks 7 k-stages per workgroup, i.e. ~4 dependent L2 / HBM round trips behind the conv kernel that
// just wrote a3 on another XCD.  Here a workgroup requests its WHOLE 32 x 448 operand slabs up front (14 coalesced
// 16-byte loads per thread, all in flight together), stages them in LDS once (57 KB, one barrier) and runs its 14 k-steps
// out of LDS: one memory round trip per launch.  (Loading the MFMA fragments straight from global memory - no LDS - was
// slower than the staged kernel: 64-byte row segments, 28 vector-memory instructions per wave: 10.5 vs 8.0 us.)
__global__ __launch_bounds__(256) void fc_act_kernel(const bf16 *__restrict__ a3, const bf16 *__restrict__ Wfc,
                                                     float *__restrict__ hpart, int M, int N) {
  constexpr int KC = FC_IN / FC_SPLITS, KS = KC / 32, VPR = KC / 8; // 448 deep: 14 MFMA k-steps, 56 vectors per row
  constexpr int ROWV = VPR + 1;                                     // LDS row pitch in 16-byte vectors (912 B: +16 B pad)
  constexpr int NV = 32 * VPR / 256;                                // 7 vectors per thread and operand
  __shared__ u32x4 sA[32 * ROWV], sB[32 * ROWV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1, z = blockIdx.z;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  u32x4 ra[NV], rb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { // vector v = tid + 256 i: row v / 56, chunk v % 56 (rows past the end: clamped, masked below)
    const int v = tid + 256 * i, r = v / VPR, c = v - r * VPR;
    ra[i] = *reinterpret_cast<const u32x4 *>(a3 + (long)min(m0 + r, M - 1) * FC_IN + z * KC + c * 8);
    rb[i] = *reinterpret_cast<const u32x4 *>(Wfc + (long)min(n0 + r, N - 1) * FC_IN + z * KC + c * 8);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + 256 * i, r = v / VPR, c = v - r * VPR;
    sA[r * ROWV + c] = ra[i];
    sB[r * ROWV + c] = rb[i];
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    Atom<bf16>::mma(sA[(wm * 16 + fr) * ROWV + ks * 4 + fg], sB[(wn * 16 + fr) * ROWV + ks * 4 + fg], acc);
  // C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
  float *out = hpart + (long)z * M * N;
  const int nn = n0 + wn * 16 + fr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int mm = m0 + wm * 16 + fg * 4 + r;
    if (mm < M && nn < N)
      out[(long)mm * N + nn] = acc[r];
  }
}
void fc_fwd_splitk(hipStream_t s, int prec, const void *a3, const void *Wfc, float *hpart, long ns, int H) {
  static const bool latency_form = tune("ALEPPO_FC_ACT_LATENCY", 1) != 0; // A/B switch
  if (prec == ALEPPO_BF16 && latency_form)
    hipLaunchKernelGGL(fc_act_kernel, grid2(ns, 32, H, 32, FC_SPLITS), dim3(256), 0, s, static_cast<const bf16 *>(a3),
                       static_cast<const bf16 *>(Wfc), hpart, (int)ns, H);
  else if (prec == ALEPPO_BF16)
    fc_fwd_splitk_t<bf16>(s, a3, Wfc, hpart, ns, H);
  else
    fc_fwd_splitk_t<float>(s, a3, Wfc, hpart, ns, H);
}

// ------------------------------------------------------------------ dgrad
template <class T>
static void fc_dgrad_t(hipStream_t s, const void *dh, const void *WfcT, const void *a3, void *dz3, long ns, int H) {
  using AL = DenseLoader<T>;
  using BL = DenseLoader<T>;
  using EP = EpiReluMask<T, 0>;
  typename AL::P ap{static_cast<const T *>(dh), H, 0};
  typename BL::P bp{static_cast<const T *>(WfcT), H, 0};
  typename EP::P ep{static_cast<T *>(dz3), static_cast<const T *>(a3), FC_IN, 0};
  static const int v = tune("ALEPPO_FC_DGRAD_TILE", 1);
  if (fc_dma() && H % Atom<T>::KT == 0) {
    const T *a = static_cast<const T *>(dh), *b = static_cast<const T *>(WfcT);
    if (v == 3)
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 64, 64, 2, 2, 16>), grid2(ns, 64, FC_IN, 64), dim3(256), 0, s, a, H,
                         b, H, ep, (int)ns, FC_IN, H);
    else if (v == 4)
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 128, 64, 2, 2, 16>), grid2(ns, 128, FC_IN, 64), dim3(256), 0, s, a,
                         H, b, H, ep, (int)ns, FC_IN, H);
    else if (v == 2)
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 128, 128, 2, 2>), grid2(ns, 128, FC_IN, 128), dim3(256), 0, s, a,
                         H, b, H, ep, (int)ns, FC_IN, H);
    else if (v == 0)
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 128, 64, 2, 2>), grid2(ns, 128, FC_IN, 64), dim3(256), 0, s, a, H,
                         b, H, ep, (int)ns, FC_IN, H);
    else
      hipLaunchKernelGGL((gemm_nt_dma_kernel<T, EP, 64, 64, 2, 2>), grid2(ns, 64, FC_IN, 64), dim3(256), 0, s, a, H, b,
                         H, ep, (int)ns, FC_IN, H);
    return;
  }
  if (v == 1)
    hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 64, 64, 2, 2>), grid2(ns, 64, FC_IN, 64), dim3(256), 0, s, ap, bp,
                       ep, (int)ns, FC_IN, H);
  else if (v == 2)
    hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 128, 2, 2>), grid2(ns, 128, FC_IN, 128), dim3(256), 0, s,
                       ap, bp, ep, (int)ns, FC_IN, H);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 64, 2, 2>), grid2(ns, 128, FC_IN, 64), dim3(256), 0, s, ap,
                       bp, ep, (int)ns, FC_IN, H);
}
template <class T>
static void conv3_dgrad_t(hipStream_t s, const void *dz3, const void *W3d, const void *a2, void *dz2, long ns) {
  using AL = DgradGatherLoader<T, 9, 9, 7, 7, 64, 3>;
  using BL = DenseLoader<T>;
  using EP = EpiReluMask<T, 0>;
  const long M = ns * 81;
  typename AL::P ap{static_cast<const T *>(dz3)};
  typename BL::P bp{static_cast<const T *>(W3d), 576, 0};
  typename EP::P ep{static_cast<T *>(dz2), static_cast<const T *>(a2), 64, 0};
  hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 64, 2, 2>), grid2(M, 128, 64, 64), dim3(256), 0, s, ap, bp,
                     ep, (int)M, 64, 576);
}
template <class T>
static void conv2_dgrad_t(hipStream_t s, const void *dz2, const void *W2d, const void *a1, void *dz1, long ns) {
  // one grid.z slice per input-pixel parity class (py,px): only the 2x2 taps with kh=py (mod 2), kw=px (mod 2)
  using AL = DgradGatherLoader<T, 10, 10, 9, 9, 64, 2>;
  using BL = DenseLoader<T>;
  using EP = EpiReluMask<T, 1>;
  const long M = ns * 100;
  typename AL::P ap{static_cast<const T *>(dz2)};
  typename BL::P bp{static_cast<const T *>(W2d), 256, 32 * 256};
  typename EP::P ep{static_cast<T *>(dz1), static_cast<const T *>(a1), 32, 0};
  hipLaunchKernelGGL((gemm_nt_kernel<T, AL, BL, EP, 128, 32, 4, 1>), grid2(M, 128, 32, 32, 4), dim3(256), 0, s, ap,
                     bp, ep, (int)M, 32, 256);
}

// ------------------------------------------------------------------ wgrad (split-K slabs)
static inline int pick_slices(long Ktot, int KP, int tiles_mn, int maxS) {
  long S = 512 / tiles_mn;
  const long cap = Ktot / (2L * KP);
  if (S > cap)
    S = cap;
  if (S > maxS)
    S = maxS;
  if (S < 1)
    S = 1;
  return (int)S;
}
static inline int chunk_for(long Ktot, int S, int KP) {
  long c = (Ktot + S - 1) / S;
  c = (c + KP - 1) / KP * KP;
  return (int)c;
}

// Split-K factor of the fc weight gradient.  2: 400 workgroups of 64 x 128 at H = 512, and two fp32 slabs (12.8 MB) for the
// reduce to read.  Measured per minibatch on one box (update loop, us): B = 4096: split 2 435.7 / 437.7, 4 438.0 / 438.0,
// 3 440.3 / 441.3, 8 444.3 / 446.8; B = 1280 (v1.yaml): split 2 232-235, split 4 241-244.
static int fc_wgrad_split_max() { return tune("ALEPPO_FC_WGRAD_SPLIT", 2); }

template <class T> static int fc_wgrad_t(hipStream_t s, const void *dh, const void *a3, float *sw, float *sb, long ns, int H) {
  using AL = DenseLoader<T>;
  using BL = DenseLoader<T>;
  constexpr int KP = Atom<T>::KT;
  static const int v = tune("ALEPPO_FC_WGRAD_TILE", 0);
  static const int smax = fc_wgrad_split_max();
  // S == 1: the "slab" IS the gradient tensor (caller passes G); S > 1: split-K slabs reduced by the caller
  const int S = (int)std::max<long>(1, std::min<long>(std::min(smax, MAXS_FC), ns / (4 * KP)));
  const int kc = chunk_for(ns, S, KP);
  typename AL::P ap{static_cast<const T *>(dh), H, 0};
  typename BL::P bp{static_cast<const T *>(a3), FC_IN, 0};
  // contiguous per-XCD work chunks (the m-tiles sharing an a3 stream on one L2) when the block count allows
  auto swz = [&](const dim3 &g) { return ((g.x * g.y * S) % 8 == 0 && tn_xcd()) ? (int)(g.x | (g.y << 12)) : 0; };
  auto grid = [&](const dim3 &g, int xsw) { return xsw ? dim3(g.x * g.y * S) : dim3(g.x, g.y, S); };
  if (v == 1) {
    const dim3 g = grid2(H, 64, FC_IN, 64);
    const int xsw = swz(g);
    hipLaunchKernelGGL((gemm_tn_kernel<T, AL, BL, 64, 64, 2, 2, true>), grid(g, xsw), dim3(256), 0, s, ap, bp, sw,
                       sb, H, FC_IN, (int)ns, kc, 1.0f, xsw);
  } else if (v == 2) {
    const dim3 g = grid2(H, 128, FC_IN, 64);
    const int xsw = swz(g);
    hipLaunchKernelGGL((gemm_tn_kernel<T, AL, BL, 128, 64, 2, 2, true>), grid(g, xsw), dim3(256), 0, s, ap, bp,
                       sw, sb, H, FC_IN, (int)ns, kc, 1.0f, xsw);
  } else {
    const dim3 g = grid2(H, 64, FC_IN, 128);
    const int xsw = swz(g);
    hipLaunchKernelGGL((gemm_tn_kernel<T, AL, BL, 64, 128, 2, 2, true>), grid(g, xsw), dim3(256), 0, s, ap, bp,
                       sw, sb, H, FC_IN, (int)ns, kc, 1.0f, xsw);
  }
  return S;
}
template <class T> static int conv3_wgrad_t(hipStream_t s, const void *dz3, const void *a2, float *sw, float *sb, long ns) {
  using AL = DenseLoader<T>;
  using BL = ConvGatherLoader<T, T, 49, 7, 1, 9, 64, 3>;
  constexpr int KP = Atom<T>::KT;
  const long K = ns * 49;
  const dim3 g = grid2(64, 64, 576, 64);
  const int S = pick_slices(K, KP, g.x * g.y, MAXS_C3);
  const int kc = chunk_for(K, S, KP);
  typename AL::P ap{static_cast<const T *>(dz3), 64, 0};
  typename BL::P bp{static_cast<const T *>(a2), 1, 81 * 64, 0, 0, 0};
  hipLaunchKernelGGL((gemm_tn_kernel<T, AL, BL, 64, 64, 2, 2, true>), dim3(g.x, g.y, S), dim3(256), 0, s, ap, bp, sw,
                     sb, 64, 576, (int)K, kc, 1.0f, 0);
  return S;
}
template <class T> static int conv2_wgrad_t(hipStream_t s, const void *dz2, const void *a1, float *sw, float *sb, long ns) {
  using AL = DenseLoader<T>;
  using BL = ConvGatherLoader<T, T, 81, 9, 2, 20, 32, 4>;
  constexpr int KP = Atom<T>::KT;
  const long K = ns * 81;
  const dim3 g = grid2(64, 64, 512, 128);
  const int S = pick_slices(K, KP, g.x * g.y, MAXS_C2);
  const int kc = chunk_for(K, S, KP);
  typename AL::P ap{static_cast<const T *>(dz2), 64, 0};
  typename BL::P bp{static_cast<const T *>(a1), 1, 400 * 32, 0, 0, 0};
  hipLaunchKernelGGL((gemm_tn_kernel<T, AL, BL, 64, 128, 2, 2, true>), dim3(g.x, g.y, S), dim3(256), 0, s, ap, bp, sw,
                     sb, 64, 512, (int)K, kc, 1.0f, 0);
  return S;
}
template <class T>
static int conv1_wgrad_t(hipStream_t s, const void *dz1, const uint32_t *obs, SampleMap map, float *sw, float *sb,
                         long ns) {
  using AL = DenseLoader<T>;
  using BL = ConvGatherLoader<T, uint8_t, 400, 20, 4, 84, 4, 8>;
  constexpr int KP = Atom<T>::KT;
  const long K = ns * 400;
  const dim3 g = grid2(32, 32, 256, 128);
  const int S = pick_slices(K, KP, g.x * g.y, MAXS_C1);
  const int kc = chunk_for(K, S, KP);
  typename AL::P ap{static_cast<const T *>(dz1), 32, 0};
  typename BL::P bp{reinterpret_cast<const uint8_t *>(obs), map.TP, map.s1 * 4, map.s0 * 4, map.base * 4, map.n0};
  hipLaunchKernelGGL((gemm_tn_kernel<T, AL, BL, 32, 128, 1, 4, true>), dim3(g.x, g.y, S), dim3(256), 0, s, ap, bp, sw,
                     sb, 32, 256, (int)K, kc, 1.0f / 255.0f, 0);
  return S;
}

// ------------------------------------------------------------------ precision dispatch
#define DISPATCH(prec, call_f32, call_bf16)                                                                            \
  do {                                                                                                                 \
    if ((prec) == ALEPPO_BF16) {                                                                                       \
      call_bf16;                                                                                                       \
    } else {                                                                                                           \
      call_f32;                                                                                                        \
    }                                                                                                                  \
  } while (0)

void conv1_fwd(hipStream_t s, int prec, const uint32_t *obs, SampleMap map, const void *W1, const float *b1, void *a1,
               long ns) {
  if (prec == ALEPPO_BF16 && use_patch_kernels())
    return patch_conv1_fwd(s, obs, map, W1, b1, a1, ns);
  DISPATCH(prec, conv1_fwd_t<float>(s, obs, map, W1, b1, a1, ns), conv1_fwd_t<bf16>(s, obs, map, W1, b1, a1, ns));
}
void conv2_fwd(hipStream_t s, int prec, const void *a1, const void *W2, const float *b2, void *a2, long ns) {
  if (prec == ALEPPO_BF16 && use_patch_kernels())
    return patch_conv2_fwd(s, a1, W2, b2, a2, ns);
  DISPATCH(prec, conv2_fwd_t<float>(s, a1, W2, b2, a2, ns), conv2_fwd_t<bf16>(s, a1, W2, b2, a2, ns));
}
void conv3_fwd(hipStream_t s, int prec, const void *a2, const void *W3, const float *b3, void *a3, long ns) {
  if (prec == ALEPPO_BF16 && use_patch_kernels())
    return patch_conv3_fwd(s, a2, W3, b3, a3, ns);
  DISPATCH(prec, conv3_fwd_t<float>(s, a2, W3, b3, a3, ns), conv3_fwd_t<bf16>(s, a2, W3, b3, a3, ns));
}
int fc_fwd(hipStream_t s, int prec, const void *a3, const void *Wfc, const float *bfc, float *h, long ns, int H,
           int max_parts) {
  if (prec == ALEPPO_BF16 && ns > 256 && H % 64 == 0 && H >= 64 && use_pipe())
    return fc_fwd_pipe(s, a3, Wfc, bfc, h, ns, H, std::max(1, std::min(max_parts, FC_FWD_MAX_PARTS)));
  DISPATCH(prec, fc_fwd_t<float>(s, a3, Wfc, bfc, h, ns, H), fc_fwd_t<bf16>(s, a3, Wfc, bfc, h, ns, H));
  return 1;
}
void fc_dgrad(hipStream_t s, int prec, const void *dh, const void *WfcT, const void *a3, void *dz3, long ns, int H) {
  if (prec == ALEPPO_BF16 && ns > 256 && H % 64 == 0 && H >= 4 * 64 && use_pipe()) // K = H: needs > NST - 1 k-stages
    return fc_dgrad_pipe(s, dh, WfcT, a3, dz3, ns, H);
  DISPATCH(prec, fc_dgrad_t<float>(s, dh, WfcT, a3, dz3, ns, H), fc_dgrad_t<bf16>(s, dh, WfcT, a3, dz3, ns, H));
}
void conv3_dgrad(hipStream_t s, int prec, const void *dz3, const void *W3d, const void *a2, void *dz2, long ns) {
  if (prec == ALEPPO_BF16 && use_patch_kernels())
    return patch_conv3_dgrad(s, dz3, W3d, a2, dz2, ns);
  DISPATCH(prec, conv3_dgrad_t<float>(s, dz3, W3d, a2, dz2, ns), conv3_dgrad_t<bf16>(s, dz3, W3d, a2, dz2, ns));
}
void conv2_dgrad(hipStream_t s, int prec, const void *dz2, const void *W2d, const void *a1, void *dz1, long ns) {
  if (prec == ALEPPO_BF16 && use_patch_kernels())
    return patch_conv2_dgrad(s, dz2, W2d, a1, dz1, ns);
  DISPATCH(prec, conv2_dgrad_t<float>(s, dz2, W2d, a1, dz1, ns), conv2_dgrad_t<bf16>(s, dz2, W2d, a1, dz1, ns));
}
int fc_wgrad_slices(int prec, long ns) {
  static const int smax = fc_wgrad_split_max();
  const int KP = prec == ALEPPO_BF16 ? Atom<bf16>::KT : Atom<float>::KT;
  return (int)std::max<long>(1, std::min<long>(std::min(smax, MAXS_FC), ns / (4 * KP)));
}
int fc_wgrad(hipStream_t s, int prec, const void *dh, const void *a3, float *sw, float *sb, long ns, int H) {
  if (prec == ALEPPO_BF16)
    return fc_wgrad_t<bf16>(s, dh, a3, sw, sb, ns, H);
  return fc_wgrad_t<float>(s, dh, a3, sw, sb, ns, H);
}
int conv3_wgrad(hipStream_t s, int prec, const void *dz3, const void *a2, float *sw, float *sb, long ns) {
  if (prec == ALEPPO_BF16 && use_patch_kernels())
    return patch_conv3_wgrad(s, dz3, a2, sw, sb, ns);
  if (prec == ALEPPO_BF16)
    return conv3_wgrad_t<bf16>(s, dz3, a2, sw, sb, ns);
  return conv3_wgrad_t<float>(s, dz3, a2, sw, sb, ns);
}
int conv2_wgrad(hipStream_t s, int prec, const void *dz2, const void *a1, float *sw, float *sb, long ns) {
  if (prec == ALEPPO_BF16 && use_patch_kernels())
    return patch_conv2_wgrad(s, dz2, a1, sw, sb, ns);
  if (prec == ALEPPO_BF16)
    return conv2_wgrad_t<bf16>(s, dz2, a1, sw, sb, ns);
  return conv2_wgrad_t<float>(s, dz2, a1, sw, sb, ns);
}
int conv1_wgrad(hipStream_t s, int prec, const void *dz1, const uint32_t *obs, SampleMap map, float *sw, float *sb,
                long ns) {
  if (prec == ALEPPO_BF16 && use_patch_kernels())
    return patch_conv1_wgrad(s, dz1, obs, map, sw, sb, ns);
  if (prec == ALEPPO_BF16)
    return conv1_wgrad_t<bf16>(s, dz1, obs, map, sw, sb, ns);
  return conv1_wgrad_t<float>(s, dz1, obs, map, sw, sb, ns);
}

} // namespace aleppo
