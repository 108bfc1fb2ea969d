// conv_fwd_fused.hpp - the update's forward convolutions in ONE launch (bf16): conv1 -> conv2 -> conv3 of a sample inside
// one workgroup, a1 and a2 handed from layer to layer through LDS.
//
// The three sample-stationary forward kernels of conv_patch.hpp are HBM-bound: per 4096-sample minibatch they
// read the observations (115 MB), write a1 (105 MB), read it back, write a2 (42 MB), read it back and write a3 (25 MB).
// Here the backward pass's copies of a1 / a2 / a3 are still WRITTEN (the dgrads' ReLU gates and the weight gradients'
// operands), but nothing is read back: 287 instead of 434 MB per minibatch, two launches fewer.
//
//   * 4 waves = ONE per SIMD, one workgroup per CU, persistent over samples: a wave may then use the whole 512-entry
//     register file, and every wave keeps TWO 16-channel atoms of all three layers resident for the whole kernel (336
//     registers; two MFMAs per LDS fragment = half the LDS reads of act_conv_kernel's one-atom waves): conv2's in VGPRs,
//     conv1's and conv3's in AGPRs (gfx950 MFMAs take their A operand from either file);
//   * hipcc schedules such a loop as read -> wait -> 2 MFMAs with the whole LDS latency exposed per fragment (the register
//     pressure heuristics win), so the hot loop is spelled out: `ds_read_b128`, `s_waitcnt lgkmcnt(N)` and the MFMAs are
//     inline assembly in a software pipeline that keeps 7 fragment reads in flight; the epilogue of pixel atom k (scale,
//     bias, ReLU, bf16, one LDS store for the next layer + one coalesced global store for the backward pass) is spread
//     over the MFMA gaps of atom k + 1;
//   * the next sample's packed uint8 stack (28 KB) is requested into 28 AGPRs per thread as soon as the current one has
//     been widened, and lands during the three phases (`s_waitcnt vmcnt(N)` counts the stores issued behind it: every
//     vector-memory operation of an iteration is unconditional - clamped indices, duplicate lanes store identical data);
//   * three barriers per sample.
#pragma once
#include "conv_patch.hpp"
#include <utility>

namespace aleppo {

struct FwdFusedParams {
  const uint32_t *obs;
  SampleMap map;
  const bf16 *w1, *w2, *w3;
  const float *b1, *b2, *b3;
  bf16 *a1, *a2, *a3;
  long ns;
};

template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F &f, std::integer_sequence<int, Is...>) {
  (f(IC<Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Two 16-channel atoms (32 output channels) of layer L for one wave, in registers for the whole kernel.  Row fr of atom a is
// output channel 32 og + (fr >> 2) * 8 + a * 4 + (fr & 3): lane (fr, fg) then leaves the two MFMAs with 8 CONSECUTIVE
// channels 32 og + 8 fg + {0..7} - one 16-byte store (conv_patch_kernel's mapping).
template <class L> struct FusedW {
  static constexpr int OGS = L::OUTC / 32, NL = 4 / OGS; // channel groups x pixel lanes = 4 waves
  static constexpr int PITCH = 32 * L::KS * 2 + 16;     // LDS image: row of K bf16 + 16 B (conflict-free fragment reads)
  u32x4 w[2][L::KS];
  float br[8];
  __device__ __forceinline__ void load_lds(const uint8_t *img, int wave, int lane) {
    const int og = wave % OGS, fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int ks = 0; ks < L::KS; ++ks)
        w[a][ks] = *reinterpret_cast<const u32x4 *>(img + (og * 32 + (fr >> 2) * 8 + a * 4 + (fr & 3)) * PITCH + ks * 64 +
                                                    fg * 16);
  }
  __device__ __forceinline__ void load_bias(const float *bias, int wave, int lane) {
    const int og = wave % OGS, fg = lane >> 4;
#pragma unroll
    for (int r = 0; r < 8; ++r)
      br[r] = bias[og * 32 + fg * 8 + r];
  }
};

// ---- the spelled-out instructions (see the header).  None of them is known to hipcc's hazard / waitcnt passes:
//  * a fragment register is written by a pending ds_read and only ever named by these statements, in program order
//    (volatile), behind the counted wait that covers it;
//  * an accumulator is read by ordinary code only after `ff_release`, which sits at least two MFMAs (32 cycles) or an
//    explicit s_nop run behind the MFMA that wrote it (MFMA result -> VALU read needs 18 wait states at most).
template <int OFF> __device__ __forceinline__ void ff_read(u32x4 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void ff_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
template <bool AGPR> __device__ __forceinline__ void ff_mfma0(f32x4 &acc, const u32x4 &w, const u32x4 &b) {
  if constexpr (AGPR)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(b));
}
template <bool AGPR> __device__ __forceinline__ void ff_mfma(f32x4 &acc, const u32x4 &w, const u32x4 &b) {
  if constexpr (AGPR)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
}
__device__ __forceinline__ void ff_release(f32x4 &a0, f32x4 &a1) { asm volatile("" : "+v"(a0), "+v"(a1)); }
__device__ __forceinline__ void ff_release_nops(f32x4 &a0, f32x4 &a1) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1));
}

__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const bf16x2 v = {(bf16)a, (bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}

template <class L> constexpr int ff_koff(int ks) { // byte offset of k-step ks inside a pixel's window of the LDS image
  constexpr int SEG = L::KW * L::C, RP = PatchGeom<L>::RP;
  const int rem = (ks * 32) % SEG;
  return 2 * (L::C >= 32 ? ((ks * 32) / SEG) * RP + (rem / L::C) * L::CP + rem % L::C : ((ks * 32) / SEG) * RP + rem);
}

// One layer of one sample out of the LDS image at byte address `in_addr`.  Every wave walks APW pixel atoms (static count;
// atoms / pixels past the end are clamped to the last pixel and recompute it: identical data to identical addresses).
// store(q, v): pixel q's 8 channels 32 og + 8 fg .. + 7 of this lane as bf16.
template <class L, bool AGPR, class Store>
__device__ __forceinline__ void fused_phase(uint32_t in_addr, const FusedW<L> &W, float scale, int wave, int lane,
                                            Store store) {
  constexpr int OGS = FusedW<L>::OGS, NL = FusedW<L>::NL, NATOM = (L::PIX + 15) / 16, KS = L::KS;
  constexpr int APW = (NATOM + NL - 1) / NL, N = APW * KS, D = 8; // D fragment registers: D - 1 reads in flight
  constexpr int RP = PatchGeom<L>::RP;
  static_assert(KS >= 8, "the epilogue of atom k is spread over k-steps 1..6 of atom k + 1");
  const int pl = wave / OGS, fr = lane & 15, fg = lane >> 4;
  uint32_t base[APW];
  int qv[APW];
#pragma unroll
  for (int ak = 0; ak < APW; ++ak) {
    const int q = min((pl + ak * NL) * 16 + fr, L::PIX - 1);
    const int oy = q / L::OW, ox = q - oy * L::OW;
    qv[ak] = q;
    base[ak] = in_addr + 2u * (uint32_t)((oy * L::S) * RP + (ox * L::S) * L::CP + fg * 8);
  }
  u32x4 frag[D];
  f32x4 acc[2][2];
  uint32_t packed[4];
  auto issue = [&](auto I) {
    constexpr int i = decltype(I)::value, ak = i / KS, ks = i % KS;
    ff_read<ff_koff<L>(ks)>(frag[i % D], base[ak]);
  };
  // epilogue of atom ak in five pieces (channels 0-1, 2-3, 4-5, 6-7, stores)
  auto epilogue = [&](auto AK, auto PIECE) {
    constexpr int ak = decltype(AK)::value, piece = decltype(PIECE)::value, set = ak & 1;
    if constexpr (piece < 4) {
      const f32x4 &a = acc[set][piece >> 1];
      constexpr int r = (piece & 1) * 2;
      packed[piece] = pack2_bf16(fmaxf(a[r] * scale + W.br[2 * piece], 0.f), fmaxf(a[r + 1] * scale + W.br[2 * piece + 1], 0.f));
    } else {
      store(qv[ak], u32x4{packed[0], packed[1], packed[2], packed[3]});
    }
  };
  static_for<D - 1>(issue);
  static_for<N>([&](auto I) {
    constexpr int i = decltype(I)::value, ak = i / KS, ks = i % KS, set = ak & 1;
    if constexpr (i + D - 1 < N)
      issue(IC<i + D - 1>{});
    ff_wait_lgkm<(N - 1 - i < D - 1 ? N - 1 - i : D - 1)>();
    if constexpr (ks == 0) {
      ff_mfma0<AGPR>(acc[set][0], W.w[0][0], frag[i % D]);
      ff_mfma0<AGPR>(acc[set][1], W.w[1][0], frag[i % D]);
    } else {
      ff_mfma<AGPR>(acc[set][0], W.w[0][ks], frag[i % D]);
      ff_mfma<AGPR>(acc[set][1], W.w[1][ks], frag[i % D]);
    }
    if constexpr (ak > 0 && ks >= 1 && ks <= 6) {
      if constexpr (ks == 1)
        ff_release(acc[set ^ 1][0], acc[set ^ 1][1]);
      else
        epilogue(IC<ak - 1>{}, IC<ks - 2>{});
    }
  });
  ff_release_nops(acc[(APW - 1) & 1][0], acc[(APW - 1) & 1][1]);
  static_for<5>([&](auto PIECE) { epilogue(IC<APW - 1>{}, PIECE); });
}

constexpr size_t FWD_FUSED_SMEM = 160 * 1024;
constexpr int FF_NT = 256; // 4 waves = one per SIMD: a wave may then use the whole 512-entry register file

// ABL: timing-only ablations (wrong results): 1 no widening, 2 / 4 / 8 no conv1 / conv2 / conv3 phase, 16 no global stores
template <int ABL>
__global__ __launch_bounds__(FF_NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void fwd_fused_kernel(FwdFusedParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  typedef __attribute__((address_space(3))) uint8_t *lds_ptr;
  bf16 *sx = reinterpret_cast<bf16 *>(smem), *s1 = sx + ACT_X_ELEMS, *s2 = s1 + ACT_A1_ELEMS;
  const uint32_t sx_addr = (uint32_t)(uintptr_t)(lds_ptr)smem, s1_addr = sx_addr + ACT_X_ELEMS * 2,
                 s2_addr = s1_addr + ACT_A1_ELEMS * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  using A1 = FusedW<LConv1Full>;
  using A2 = FusedW<LConv2FwdSmall>;
  using A3 = FusedW<LConv3FwdSmall>;
  static_assert(64 * A3::PITCH <= FWD_FUSED_SMEM && 64 * A2::PITCH <= FWD_FUSED_SMEM, "weight images");
  static_assert(ACT_ACT_BYTES <= FWD_FUSED_SMEM, "activation images");
  constexpr int RP2 = PatchGeom<LConv2Fwd>::RP, RP3 = PatchGeom<LConv3Fwd>::RP;
  constexpr int NXV = 1764, NR = (NXV + FF_NT - 1) / FF_NT; // packed stack: 16-byte vectors, per thread
  // global stores per sample and thread behind the stack prefetch: one per pixel atom of the three phases
  constexpr int NSTORES = (ABL & 16) ? 0 : ((ABL & 2) ? 0 : 7) + ((ABL & 4) ? 0 : 3) + ((ABL & 8) ? 0 : 2);
  static_assert(NR == 7, "the prefetch wait names R[0..6]");
  A1 W1;
  A2 W2;
  A3 W3;
  u32x4 R[NR]; // AGPRs (written by the inline-assembly loads below)
  const long gs = gridDim.x;
  long n = blockIdx.x;
  if (n >= P.ns)
    return;
  auto load_obs = [&](long m) { // unconditional: sample and vector index clamped
    const long nn = min(m, P.ns - 1) + P.map.n0;
    const long off = (nn / P.map.TP) * P.map.s1 + (nn % P.map.TP) * P.map.s0 + P.map.base;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(P.obs + off);
#pragma unroll
    for (int i = 0; i < NR; ++i)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(R[i]) : "v"(src + min(tid + FF_NT * i, NXV - 1)) : "memory");
  };
  load_obs(n); // (first: its round trip to HBM runs under the weight set-up below)
  { // ---- the three layers' weights: one coalesced copy per workgroup -> padded LDS image -> this wave's two atoms
    auto img_off = [](int v, int vpr, int pitch) { return (v / vpr) * pitch + (v % vpr) * 16; };
    W1.load_bias(P.b1, wave, lane);
    W2.load_bias(P.b2, wave, lane);
    W3.load_bias(P.b3, wave, lane);
    u32x4 S[18];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      S[i] = reinterpret_cast<const u32x4 *>(P.w1)[tid + FF_NT * i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<u32x4 *>(smem + img_off(tid + FF_NT * i, 32, A1::PITCH)) = S[i];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      S[i] = reinterpret_cast<const u32x4 *>(P.w2)[tid + FF_NT * i];
    __syncthreads();
    W1.load_lds(smem, wave, lane);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i)
      *reinterpret_cast<u32x4 *>(smem + img_off(tid + FF_NT * i, 64, A2::PITCH)) = S[i];
#pragma unroll
    for (int i = 0; i < 18; ++i)
      S[i] = reinterpret_cast<const u32x4 *>(P.w3)[tid + FF_NT * i];
    __syncthreads();
    W2.load_lds(smem, wave, lane);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 18; ++i)
      *reinterpret_cast<u32x4 *>(smem + img_off(tid + FF_NT * i, 72, A3::PITCH)) = S[i];
    __syncthreads();
    W3.load_lds(smem, wave, lane);
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)"
               : "+a"(R[0]), "+a"(R[1]), "+a"(R[2]), "+a"(R[3]), "+a"(R[4]), "+a"(R[5]), "+a"(R[6])::"memory");
  auto pk = [](uint32_t lo, uint32_t hi) { return pack_u8_pair_bf16(lo, hi); };
  const int og2 = wave % 2, fg = lane >> 4;
  for (; n < P.ns; n += gs) {
    // ---- widen the packed stack (R) into sx; then ask for the next sample's
    if constexpr (!(ABL & 1))
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int v = tid + FF_NT * i;
      if (v < NXV) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const uint32_t w0 = R[i][2 * d], w1 = R[i][2 * d + 1];
          reinterpret_cast<u32x4 *>(sx)[2 * v + d] =
              u32x4{pk(w0 & 255u, (w0 >> 8) & 255u), pk((w0 >> 16) & 255u, w0 >> 24),
                    pk(w1 & 255u, (w1 >> 8) & 255u), pk((w1 >> 16) & 255u, w1 >> 24)};
        }
      }
    }
    __syncthreads(); // (also: every wave is done with conv3 of the previous sample - s2 may be rewritten two barriers on)
    load_obs(n + gs);
    if constexpr (!(ABL & 2)) {
      bf16 *g1 = P.a1 + n * (long)(400 * 32) + fg * 8;
      fused_phase<LConv1Full, true>(sx_addr, W1, 1.0f / 255.0f, wave, lane, [&](int q, u32x4 v) {
        *reinterpret_cast<u32x4 *>(s1 + (q / 20) * RP2 + (q % 20) * LConv2Fwd::CP + fg * 8) = v;
        if constexpr (!(ABL & 16))
        *reinterpret_cast<u32x4 *>(g1 + q * 32) = v;
      });
    }
    __syncthreads();
    if constexpr (!(ABL & 4)) {
      bf16 *g2 = P.a2 + n * (long)(81 * 64) + og2 * 32 + fg * 8;
      fused_phase<LConv2FwdSmall, false>(s1_addr, W2, 1.0f, wave, lane, [&](int q, u32x4 v) {
        *reinterpret_cast<u32x4 *>(s2 + (q / 9) * RP3 + (q % 9) * LConv3Fwd::CP + og2 * 32 + fg * 8) = v;
        if constexpr (!(ABL & 16))
        *reinterpret_cast<u32x4 *>(g2 + q * 64) = v;
      });
    }
    __syncthreads();
    if constexpr (!(ABL & 8)) {
      bf16 *g3 = P.a3 + n * (long)(49 * 64) + og2 * 32 + fg * 8;
      fused_phase<LConv3FwdSmall, true>(s2_addr, W3, 1.0f, wave, lane,
                                        [&](int q, u32x4 v) { if constexpr (!(ABL & 16)) *reinterpret_cast<u32x4 *>(g3 + q * 64) = v; });
    }
    // the prefetched stack: every vector-memory operation issued behind it is one of the NSTORES stores above
    asm volatile("s_waitcnt vmcnt(%7)"
                 : "+a"(R[0]), "+a"(R[1]), "+a"(R[2]), "+a"(R[3]), "+a"(R[4]), "+a"(R[5]), "+a"(R[6])
                 : "n"(NSTORES)
                 : "memory");
  }
}

} // namespace aleppo
