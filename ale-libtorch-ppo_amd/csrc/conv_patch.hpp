// conv_patch.hpp - sample-stationary bf16 convolution kernels for gfx950 (the performance path).
//
// The generic gather-GEMMs of gemm.hpp re-stage an im2col tile per 64-deep k-step and pay a full
// prologue per 128-row tile - fine for parity (fp32), wasteful for these skinny problems (N = 32/64
// output channels, K = 256..576).  Here each workgroup is PERSISTENT (one per CU, 8 waves) and streams
// samples: the input patch of SB samples (<= 62 KB as bf16) is staged ONCE in LDS (conv1: u8 -> bf16
// widened once instead of once per 8x8 window, 4x less conversion work), the layer's weights live in
// REGISTERS for the whole kernel (each wave owns two 16-channel atoms = 64..144 VGPRs), and the MFMA
// operands are read from the patch with im2col addressing (base + immediate offset per k-step).
// Next group's global loads are in flight during the current group's MFMAs (register prefetch, LDS double
// buffer, one barrier per group).
//
//   conv_patch_kernel<L>  forward (conv1/2/3) and dgrad (conv3, conv2 by parity class):
//        D[oc][pixel] = sum_k W[oc][k] * gather(patch)[pixel][k]        (A = weights, B = patch)
//   conv_wgrad_patch_kernel<L>: dW[oc][j] += sum_pixel dY[pixel][oc] * im2col(X)[pixel][j]
//        accumulators stay in registers across ALL samples of the workgroup (one slab per workgroup);
//        both operands are k(=pixel)-outer in LDS and read with ds_read_b64_tr_b16.
#pragma once
#include "gemm.hpp"
#include <type_traits>

namespace aleppo {

enum PatchMode { PM_FWD = 0, PM_DGRAD = 1 };

// LDS pixel pitch CP: a patch pixel's C channels are followed by CP-C pad elements.  With the natural pitch
// (64 or 128 B) the 16 lanes of an operand read (consecutive output pixels) fall on 2 bank groups - measured
// 60-73 % of LDS cycles were conflict cycles; CP = 40 / 72 / 80 spreads them over 8-16 distinct 16/32-B slots.
// ---- layer descriptors ---------------------------------------------------------------------------
// forward: NHWC input [IH][IW][C], KHxKW window, stride S -> PIX = OHxOW output pixels, OUTC channels
// conv1 works on HALF samples (output rows 0-9 / 10-19 <- input rows 0-43 / 40-83) so that two LDS buffers
// of the widened patch stay small (2 x 29.6 KB): GPS = groups per sample, GSTRIDE = source elements between
// the starts of consecutive groups of one sample.
struct LConv1Fwd {
  static constexpr int MODE = PM_FWD;
  using InT = uint8_t;
  static constexpr int IN_ELEMS = 44 * 84 * 4, PIX = 200, OW = 20, S = 4, IW = 84, C = 4, KW = 8, OUTC = 32, KS = 8,
                       SB = 1, OG = 1, CLASSES = 1, GPS = 2, GSTRIDE = 40 * 84 * 4, CP = 4, RPAD = 0, SPAD = 0, STATIC_ATOMS = 1, ONE_ATOM = 0;
};
struct LConv2Fwd {
  static constexpr int MODE = PM_FWD;
  using InT = bf16;
  static constexpr int IN_ELEMS = 400 * 32, PIX = 81, OW = 9, S = 2, IW = 20, C = 32, KW = 4, OUTC = 64, KS = 16,
                       SB = 2, OG = 2, CLASSES = 1, GPS = 1, GSTRIDE = 0, CP = 40, RPAD = 8, SPAD = 0, STATIC_ATOMS = 1, ONE_ATOM = 0;
};
struct LConv3Fwd {
  static constexpr int MODE = PM_FWD;
  using InT = bf16;
  static constexpr int IN_ELEMS = 81 * 64, PIX = 49, OW = 7, S = 1, IW = 9, C = 64, KW = 3, OUTC = 64, KS = 18, SB = 6,
                       OG = 2, CLASSES = 1, GPS = 1, GSTRIDE = 0, CP = 80, RPAD = 96, SPAD = 32, STATIC_ATOMS = 0, ONE_ATOM = 0;
};
// acting-size variants (ns <= 256): one sample per group so that every CU gets a workgroup
struct LConv2FwdSmall : LConv2Fwd {
  static constexpr int SB = 1;
};
struct LConv3FwdSmall : LConv3Fwd {
  static constexpr int SB = 1;
};
// dgrad: input = dY [OH][OW][OCK], output pixel grid PH x PW, taps TH x TW, source pixel (y-dy, x-dx)
struct LConv3Dgrad {
  static constexpr int MODE = PM_DGRAD;
  using InT = bf16;
  static constexpr int IN_ELEMS = 49 * 64, PIX = 81, PW = 9, OH = 7, OW = 7, OCK = 64, TW = 3, OUTC = 64, KS = 18,
                       SB = 8, OG = 2, CLASSES = 1, GPS = 1, GSTRIDE = 0, C = 64, CP = 72, RPAD = 80, SPAD = 0,
                       PRELOAD_GATES = 0, ONE_ATOM = 0;
};
struct LConv2Dgrad { // one parity class (py,px) of the 20x20 input per wave group; 2x2 live taps
  static constexpr int MODE = PM_DGRAD;
  using InT = bf16;
  static constexpr int IN_ELEMS = 81 * 64, PIX = 100, PW = 10, OH = 9, OW = 9, OCK = 64, TW = 2, OUTC = 32, KS = 8,
                       SB = 2, OG = 4, CLASSES = 4, GPS = 1, GSTRIDE = 0, C = 64, CP = 80, RPAD = 80, SPAD = 0,
                       PRELOAD_GATES = 0, ONE_ATOM = 0;
};

// LDS image of one unit: pixel (row, col) of the source at row*RP + col*CP bf16 elements, units SP apart.  The pixel
// pitch CP, the row padding RPAD and the unit padding SPAD are chosen (tests/tools/lds_conflicts.py simulates the
// gfx950 ds_read_b128 lane groups) so that the 16 lanes of a group hit 16 different 16-byte slots for almost every
// (atom, k-step): 4.1-5.3 LDS cycles per fragment read instead of 7.4-11.2 with a plain padded pixel pitch.
template <class L> constexpr int patch_src_width() { // source row width in pixels
  if constexpr (L::MODE == PM_FWD)
    return L::IW;
  else
    return L::OW;
}
// dgrad descriptors may ask for the ReLU gates of ALL atoms of a group to be requested before its first MFMA
// (PRELOAD_GATES = 1) and the atom loop to be unrolled: the loop is then free of loads, so hipcc no longer drains
// the vector-memory queue (the previous atom's store + this atom's gate, a full memory round trip) once per atom.
// conv2 dgrad: 90 -> 55 us at HALF the occupancy (196 VGPRs, 2 workgroups per CU).  conv3 dgrad (18 k-steps of
// weights in registers) has no room for it: 2 samples per group measured equal, 4 spill.
// forward descriptors opt in to the static atom loop with STATIC_ATOMS = 1 (conv2 fwd: 41 -> 37 us, conv1 fwd: 61 ->
// 55 us with ONE register set - a second set costs it a wave of occupancy; conv3 fwd gets slower and stays dynamic)
template <class L> constexpr bool static_atoms() {
  if constexpr (L::MODE == PM_FWD)
    return L::STATIC_ATOMS != 0;
  else
    return false;
}
template <class L> constexpr bool preload_gates() {
  if constexpr (L::MODE == PM_FWD)
    return false;
  else
    return L::PRELOAD_GATES != 0;
}
template <class L> struct PatchGeom {
  static constexpr int IWL = patch_src_width<L>();
  static constexpr int ROWS = L::IN_ELEMS / L::C / IWL;
  static constexpr int RP = IWL * L::CP + L::RPAD;
  static constexpr int SP = ROWS * RP + L::SPAD;
};

struct PatchParams {
  const void *in;     // activations / dY (bf16) or packed u8 stacks
  const bf16 *w;      // [CLASSES][OUTC][32*KS]
  const float *bias;  // fwd
  const bf16 *act;    // dgrad: activation whose ReLU gates the result (same indexing as out)
  bf16 *out;
  long ns;            // samples
  SampleMap map;      // conv1 only: where sample n's packed stack lives (units: u32 pixels)
  float scale;        // fwd epilogue: v*scale + bias
  bf16 *dummy;        // fwd: >= 8 KB scratch that lanes without an output pixel store to (see the atom loop)
};

__device__ __forceinline__ u32x2 pack4_bf16(float a, float b, float c, float d) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const bf16x2 lo = {(bf16)a, (bf16)b}, hi = {(bf16)c, (bf16)d};
  return u32x2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __uint_as_float(hi16 << 16); }

template <class L, int NW> __global__ __launch_bounds__(64 * NW) void conv_patch_kernel(PatchParams P) {
  using InT = typename L::InT;
  constexpr bool U8 = sizeof(InT) == 1;
  constexpr int K = 32 * L::KS;
  constexpr int PATCH = L::IN_ELEMS;                         // source elements per unit
  using GEO = PatchGeom<L>;
  constexpr int LPATCH = GEO::SP, RP = GEO::RP;              // bf16 elements per unit / per source row in LDS
  constexpr int BUF_ELEMS = (L::SB * LPATCH + 63) / 64 * 64; // per buffer
  constexpr int SRC_VECS = L::SB * PATCH * (int)sizeof(InT) / 16; // 16-byte source vectors per group
  constexpr int NT = 64 * NW;                                // threads per workgroup
  constexpr int NV = (SRC_VECS + NT - 1) / NT;
  constexpr int NL = NW / L::OG;                             // pixel lanes (waves sharing an oc group)
  static_assert(NW % L::OG == 0, "waves per workgroup must be a multiple of the channel groups");
  constexpr int NATOM = (L::SB * L::PIX + 15) / 16;
  static_assert((L::SB * PATCH * (int)sizeof(InT)) % 16 == 0, "group size");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16 *sbuf = reinterpret_cast<bf16 *>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int og = wave % L::OG, pl = wave / L::OG;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- weights of this wave's two 16-channel atoms -> registers (A operand: row = fr, k = 32ks + 8fg)
  // NA = 16-channel MFMA atoms per wave.  Two atoms share every LDS fragment (half the LDS reads per MFMA) but their
  // weights fill the register file for the 18-k-step layers; ONE_ATOM descriptors keep one atom per wave (4 channel
  // groups of 16) so that the compiler can hold many fragments in flight instead of a read -> MFMA -> read chain.
  constexpr int NA = L::ONE_ATOM ? 1 : 2, CPW = 16 * NA;
  const int wrow0 = (L::CLASSES > 1 ? og * L::OUTC : og * CPW);
  u32x4 W[NA][L::KS];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int ks = 0; ks < L::KS; ++ks)
      // Row fr of atom a is output channel (fr >> 2) * 8 + a * 4 + (fr & 3) of this wave's 32: the MFMA result then
      // leaves lane (fr, fg) with channels fg * 8 + {0..3} (atom 0) and fg * 8 + {4..7} (atom 1) - 8 CONSECUTIVE
      // channels, so every epilogue access is one 16-byte piece and the 4 lanes of a pixel cover 64 contiguous bytes
      // (8-byte pieces at a 32-byte stride cost conv2 dgrad 40 of its 100 us in partial-line stores).
      W[a][ks] = *reinterpret_cast<const u32x4 *>(
          P.w + (long)(wrow0 + (NA == 2 ? (fr >> 2) * 8 + a * 4 + (fr & 3) : fr)) * K + ks * 32 + fg * 8);
  const int oc0 = (L::CLASSES > 1 ? 0 : og * CPW) + fg * 4 * NA; // first of this lane's 4 NA consecutive output channels
  float bias_r[NA][4];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      bias_r[a][r] = (L::MODE == PM_FWD) ? P.bias[oc0 + a * 4 + r] : 0.f;

  // a "unit" is a sample (GPS == 1) or a half sample (conv1); a group is SB consecutive units
  const long nunits = P.ns * L::GPS;
  const long ngroups = (nunits + L::SB - 1) / L::SB;
  // Prefetch registers.  Loads are UNCONDITIONAL (vector index and group clamped, units past the end zeroed when they
  // are staged): a predicated load may be skipped by a whole wave, and then hipcc cannot count the outstanding
  // vector-memory operations and drains them all (vmcnt(0)) at the next use.
  struct Regs {
    u32x4 r[NV];
  };
  Regs R0;
  auto gload = [&](Regs &RR, long grp) {
    u32x4 (&R)[NV] = RR.r;
    if constexpr (static_atoms<L>()) {
      const long n0 = min(grp, ngroups - 1) * L::SB;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = min(tid + NT * i, SRC_VECS - 1);
        if constexpr (U8) { // SB == 1: one half of a packed stack per group, located through the slot map
          const long n = n0 / L::GPS + P.map.n0;
          const long off = (n / P.map.TP) * P.map.s1 + (n % P.map.TP) * P.map.s0 + P.map.base; // u32 pixels
          R[i] = reinterpret_cast<const u32x4 *>(static_cast<const uint8_t *>(P.in) + off * 4 +
                                                 (n0 % L::GPS) * (long)L::GSTRIDE)[v];
        } else {
          const long u = min(n0 + v / (PATCH / 8), nunits - 1);
          R[i] = reinterpret_cast<const u32x4 *>(static_cast<const bf16 *>(P.in) + u * PATCH)[v % (PATCH / 8)];
        }
      }
    } else { // predicated loads (the kernels whose atom loop is dynamic drain the queue at the staging write anyway)
      const long n0 = grp * L::SB;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = tid + NT * i;
        R[i] = zero16();
        if (v < SRC_VECS && grp < ngroups) {
          if constexpr (U8) {
            const long n = n0 / L::GPS + P.map.n0;
            const long off = (n / P.map.TP) * P.map.s1 + (n % P.map.TP) * P.map.s0 + P.map.base; // u32 pixels
            R[i] = reinterpret_cast<const u32x4 *>(static_cast<const uint8_t *>(P.in) + off * 4 +
                                                   (n0 % L::GPS) * (long)L::GSTRIDE)[v];
          } else {
            const int s = v / (PATCH / 8);
            if (n0 + s < nunits)
              R[i] = reinterpret_cast<const u32x4 *>(static_cast<const bf16 *>(P.in) + n0 * PATCH)[v];
          }
        }
      }
    }
  };
  auto swrite = [&](Regs &RR, int buf, long grp) {
    u32x4 (&R)[NV] = RR.r;
    bf16 *dst = sbuf + (size_t)buf * BUF_ELEMS;
    const long u0 = grp * L::SB;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + NT * i;
      if (v < SRC_VECS) {
        if constexpr (static_atoms<L>())
          if (u0 + v / (SRC_VECS / L::SB) >= nunits) // unit past the end: stage zeros
            R[i] = zero16();
        if constexpr (U8) { // 16 bytes -> 16 bf16 (exact), two LDS vectors
          auto pk = [](uint32_t lo, uint32_t hi) {
            return pack_u8_pair_bf16(lo, hi);
          };
          u32x4 o0, o1;
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const uint32_t w0 = R[i][2 * d], w1 = R[i][2 * d + 1];
            const u32x4 o = {pk(w0 & 255u, (w0 >> 8) & 255u), pk((w0 >> 16) & 255u, w0 >> 24),
                             pk(w1 & 255u, (w1 >> 8) & 255u), pk((w1 >> 16) & 255u, w1 >> 24)};
            if (d == 0)
              o0 = o;
            else
              o1 = o;
          }
          reinterpret_cast<u32x4 *>(dst)[2 * v] = o0;
          reinterpret_cast<u32x4 *>(dst)[2 * v + 1] = o1;
        } else { // vector v = 8 channels of pixel (v / VPP) % PIXIN of unit v / (VPP * PIXIN)
          constexpr int VPP = L::C / 8, PIXIN = PATCH / L::C;
          const int px = v / VPP, u = px / PIXIN, pp = px - u * PIXIN, row = pp / GEO::IWL, col = pp - row * GEO::IWL;
          *reinterpret_cast<u32x4 *>(dst + u * LPATCH + row * RP + col * L::CP + (v % VPP) * 8) = R[i];
        }
      }
    }
  };

  constexpr bool STATIC = static_atoms<L>(); // static atom count + unconditional stores (see process())
  auto process = [&](int bufi, long grp) {
    const bf16 *pb = sbuf + (size_t)bufi * BUF_ELEMS;
    const long n0 = grp * L::SB; // first unit of this group; unit u covers output pixels [u*PIX, (u+1)*PIX)
    const int count = (int)max(0L, min((long)L::SB, nunits - n0));
    constexpr int APW = (NATOM + NL - 1) / NL; // atoms per wave and group
    constexpr bool PRE = L::MODE != PM_FWD && preload_gates<L>();
    u32x4 gates[PRE ? APW : 1];
    if constexpr (PRE) {
#pragma unroll
      for (int ak = 0; ak < APW; ++ak) {
        const int a_ = pl + ak * NL, q_ = a_ * 16 + fr;
        const int s_ = min(q_ / L::PIX, L::SB - 1), p_ = q_ - (q_ / L::PIX) * L::PIX;
        const int y_ = p_ / L::PW, x_ = p_ - y_ * L::PW;
        long off_;
        if constexpr (L::CLASSES > 1)
          off_ = (((n0 + s_) * 20 + 2 * y_ + (og >> 1)) * 20 + 2 * x_ + (og & 1)) * (long)L::OUTC + oc0;
        else
          off_ = ((n0 + s_) * L::PIX + p_) * (long)L::OUTC + oc0;
        gates[ak] = zero16();
        if (a_ < NATOM && q_ < count * L::PIX)
          gates[ak] = *reinterpret_cast<const u32x4 *>(P.act + off_);
      }
    }
    auto atom_body = [&](int atom, u32x4 gate_pre) {
      const int q = atom * 16 + fr;
      const bool qok = atom < NATOM && q < count * L::PIX;
      const int s = min(q / L::PIX, L::SB - 1), p = q - (q / L::PIX) * L::PIX;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      auto mma2 = [&](int ks, const u32x4 &b) {
        Atom<bf16>::mma(W[0][ks], b, acc0);
        if constexpr (NA == 2)
          Atom<bf16>::mma(W[NA - 1][ks], b, acc1);
      };
      long out_off;
      u32x4 gate = gate_pre;
      if constexpr (L::MODE == PM_FWD) {
        const int oy = p / L::OW, ox = p - oy * L::OW;
        const int base = qok ? s * LPATCH + (oy * L::S) * RP + (ox * L::S) * L::CP + fg * 8 : fg * 8;
        constexpr int SEG = L::KW * L::C;
#pragma unroll
        for (int ks = 0; ks < L::KS; ++ks) {
          // k = 32ks + 8fg + j  ->  (kh, kw, c):  kh = 32ks / SEG, kw*C + c = 32ks % SEG (+ 8fg + j)
          const int rem = (ks * 32) % SEG;
          const int koff = L::C >= 32 ? ((ks * 32) / SEG) * RP + (rem / L::C) * L::CP + rem % L::C
                                      : ((ks * 32) / SEG) * RP + rem;
          const u32x4 b = *reinterpret_cast<const u32x4 *>(pb + base + koff);
          mma2(ks, b);
        }
        out_off = ((n0 + s) * L::PIX + p) * (long)L::OUTC + oc0;
      } else {
        const int y = p / L::PW, x = p - y * L::PW;
        if constexpr (L::CLASSES > 1) { // conv2 dgrad: pixel (2y+py, 2x+px) of the 20x20x32 tensor
          const int py = og >> 1, px = og & 1;
          out_off = (((n0 + s) * 20 + 2 * y + py) * 20 + 2 * x + px) * (long)L::OUTC + oc0;
        } else {
          out_off = ((n0 + s) * L::PIX + p) * (long)L::OUTC + oc0;
        }
        // the ReLU gate is fetched BEFORE the MFMA chain: issued in the epilogue its full global-memory latency
        // was exposed once per atom (PMC: 67 % of conv2 dgrad's wave cycles parked in s_waitcnt)
        if constexpr (!PRE)
          if (qok) {
            if constexpr (NA == 2) {
              gate = *reinterpret_cast<const u32x4 *>(P.act + out_off);
            } else {
              const u32x2 g2 = *reinterpret_cast<const u32x2 *>(P.act + out_off);
              gate = u32x4{g2[0], g2[1], 0u, 0u};
            }
          }
        const int base = s * LPATCH + fg * 8;
        constexpr int KPT = L::OCK / 32; // k-steps per tap
#pragma unroll
        for (int ks = 0; ks < L::KS; ++ks) {
          const int tap = ks / KPT, dy = tap / L::TW, dx = tap - dy * L::TW;
          const int sy = y - dy, sx = x - dx;
          const bool ok = qok && sy >= 0 && sy < L::OH && sx >= 0 && sx < L::OW;
          // branch-free: always read, then zero invalid taps -> the compiler can issue all KS LDS reads ahead
          // of the MFMAs.  Invalid taps read the CLAMPED source pixel: the address then equals a neighbour
          // lane's (broadcast) instead of piling every invalid lane onto one bank.
          const int cy = min(max(sy, 0), L::OH - 1), cx = min(max(sx, 0), L::OW - 1);
          const int off = base + cy * RP + cx * L::CP + (ks % KPT) * 32;
          u32x4 b = *reinterpret_cast<const u32x4 *>(pb + off);
          b = ok ? b : zero16();
          mma2(ks, b);
        }
      }
      if constexpr (L::MODE == PM_FWD) {
        // EVERY lane stores (lanes without a pixel to the scratch `dummy`), and every wave runs the same number of
        // atoms: the vector-memory operations per group are then a compile-time constant, hipcc waits for the
        // prefetched registers with a COUNTED vmcnt and these stores are never drained inside the loop.
        float v0[4], v1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v0[r] = fmaxf(acc0[r] * P.scale + bias_r[0][r], 0.f);
          v1[r] = fmaxf(acc1[r] * P.scale + bias_r[NA - 1][r], 0.f);
        }
        const u32x2 lo = pack4_bf16(v0[0], v0[1], v0[2], v0[3]), hi = pack4_bf16(v1[0], v1[1], v1[2], v1[3]);
        bf16 *dstp = P.out + out_off;
        if constexpr (STATIC)
          dstp = qok ? dstp : P.dummy + tid * 8;
        if (STATIC || qok) {
          if constexpr (NA == 2)
            *reinterpret_cast<u32x4 *>(dstp) = u32x4{lo[0], lo[1], hi[0], hi[1]};
          else
            *reinterpret_cast<u32x2 *>(dstp) = lo;
        }
      } else if (qok) {
        {
          auto gated = [](uint32_t w, int h, float v) { // ReLU gate on the stored bf16 activation
            const float a = bf16_bits_to_f32(h ? (w >> 16) : (w & 0xFFFFu));
            return a > 0.f ? v : 0.f;
          };
          const u32x2 lo = pack4_bf16(gated(gate[0], 0, acc0[0]), gated(gate[0], 1, acc0[1]), gated(gate[1], 0, acc0[2]),
                                      gated(gate[1], 1, acc0[3]));
          const u32x2 hi = pack4_bf16(gated(gate[2], 0, acc1[0]), gated(gate[2], 1, acc1[1]), gated(gate[3], 0, acc1[2]),
                                      gated(gate[3], 1, acc1[3]));
          if constexpr (NA == 2)
            *reinterpret_cast<u32x4 *>(P.out + out_off) = u32x4{lo[0], lo[1], hi[0], hi[1]};
          else
            *reinterpret_cast<u32x2 *>(P.out + out_off) = lo;
        }
      }
    };
    if constexpr (PRE) {
#pragma unroll
      for (int ak = 0; ak < APW; ++ak)
        if (pl + ak * NL < NATOM)
          atom_body(pl + ak * NL, gates[ak]);
    } else if constexpr (STATIC) {
#pragma unroll
      for (int ak = 0; ak < APW; ++ak)
        atom_body(pl + ak * NL, zero16());
    } else {
      for (int atom = pl; atom < NATOM; atom += NL)
        atom_body(atom, zero16());
    }
  };

  const long gs = gridDim.x;
  long grp = blockIdx.x;
  // one group ahead, ONE register set (a second set - two groups in flight - cost conv1 fwd a wave of occupancy: 66 vs
  // 55 us; section 4 of DESIGN.md)
  gload(R0, grp);
  swrite(R0, 0, grp);
  gload(R0, grp + gs);
  __syncthreads();
  for (int it = 0; grp < ngroups; grp += gs, ++it) {
    process(it & 1, grp);
    swrite(R0, (it + 1) & 1, grp + gs);
    gload(R0, grp + 2 * gs);
    __syncthreads();
  }
}

// ================================================================================================
// Acting path: conv1 -> conv2 -> conv3 of ONE sample per workgroup in a single launch.  The widened
// 84x84x4 stack (56 KB), a1 (25.6 KB) and a2 (10.4 KB) live in LDS; only a3 (6 KB) goes back to HBM for
// the batched fc GEMM.  At E_g = 128 envs this replaces three latency-bound launches per rollout slot.
// ================================================================================================
struct LConv1Full : LConv1Fwd {
  static constexpr int IN_ELEMS = 84 * 84 * 4, PIX = 400, GPS = 1, GSTRIDE = 0;
};
struct ActConvParams {
  const uint32_t *obs;
  SampleMap map;
  const bf16 *w1, *w2, *w3;
  const float *b1, *b2, *b3;
  bf16 *a3;
  long ns;
};
// Frame ingest fused in front of the acting convolutions (MODE 1 / 2 of act_conv_kernel): the workgroup of environment
// n first forms the stack it is about to act on - new frame (MODE 1: given 84x84 bytes; MODE 2: palette LUT + 84x84
// area resize + 2-frame max of a raw 210x160 pair, the arithmetic of ingest_kernel<true>) shifted into / broadcast
// over the previous slot's packed stack (rollout.cc:184-196) - writes it to its rollout slot (Buffer::add's
// observation copy, buffer.cc:47) and keeps it in registers for the widening step: the stack never makes the
// HBM round trip between an ingest launch and a convolution launch, and the slot's critical path loses a launch.
struct ActIngestParams {
  const uint8_t *frames; // MODE 1: [ns][84*84]; MODE 2: [ns][2][210][160]; device or mapped host memory
  const uint8_t *lut;    // MODE 2: 256-entry palette -> gray table
  uint32_t *obs_rw;      // the observation slots (same array as ActConvParams::obs)
  long src_delta;        // previous slot - acted slot, in u32 pixels (the stack to shift)
  StartBits sbits;       // episode-start flags (bit e of word e / 32)
  const uint8_t *start_bytes; // != nullptr: the flags as bytes in (mapped host) memory instead - a step enqueued before
                              // the emulator has produced them (aleppo_arm_step)
};

// Acting phases use ONE 16-channel atom per wave (8 waves = MA channel atoms x 8/MA pixel lanes): the weights
// of all three layers then fit in registers together (8 + 16 + 18 fragments = 168 VGPRs) and are loaded once,
// at kernel start, overlapped with staging the observation - no exposed weight round trip between phases.
template <class L> struct ActW {
  static constexpr int MA = L::OUTC / 16, NL = 8 / MA;
  u32x4 w[L::KS];
  float br[4];
  __device__ __forceinline__ void load(const bf16 *wp, const float *bias, int wave, int lane) {
    const int og = wave % MA, fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < L::KS; ++ks)
      w[ks] = *reinterpret_cast<const u32x4 *>(wp + (long)(og * 16 + fr) * (32 * L::KS) + ks * 32 + fg * 8);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      br[r] = bias[og * 16 + fg * 4 + r];
  }
  // this wave's atom out of an LDS image of the layer's weights: row r at img + r * PITCH bytes (PITCH = row + 16 B)
  static constexpr int PITCH = 32 * L::KS * 2 + 16;
  __device__ __forceinline__ void load_lds(const uint8_t *rows16, int lane) { // rows16: first of the atom's 16 rows
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < L::KS; ++ks)
      w[ks] = *reinterpret_cast<const u32x4 *>(rows16 + fr * PITCH + ks * 64 + fg * 16);
  }
  __device__ __forceinline__ void load_bias(const float *bias, int wave, int lane) {
    const int og = wave % MA, fg = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      br[r] = bias[og * 16 + fg * 4 + r];
  }
};
// one forward phase out of LDS patch `pb`; writes 4 channels x 1 pixel per lane through `store(q, oc, v4)`
template <class L, class Store>
__device__ __forceinline__ void act_phase(const bf16 *pb, const ActW<L> &W, float scale, int wave, int lane,
                                          Store store) {
  constexpr int MA = ActW<L>::MA, NL = ActW<L>::NL, NATOM = (L::PIX + 15) / 16, SEG = L::KW * L::C;
  constexpr int RP = PatchGeom<L>::RP;
  const int og = wave % MA, pl = wave / MA, fr = lane & 15, fg = lane >> 4;
  const int oc0 = og * 16 + fg * 4;
  for (int atom = pl; atom < NATOM; atom += NL) {
    const int q = atom * 16 + fr;
    const bool qok = q < L::PIX;
    const int oy = q / L::OW, ox = q - oy * L::OW;
    const int base = qok ? (oy * L::S) * RP + (ox * L::S) * L::CP + fg * 8 : fg * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < L::KS; ++ks) {
      const int rem = (ks * 32) % SEG;
      const int koff = L::C >= 32 ? ((ks * 32) / SEG) * RP + (rem / L::C) * L::CP + rem % L::C
                                  : ((ks * 32) / SEG) * RP + rem;
      const u32x4 b = *reinterpret_cast<const u32x4 *>(pb + base + koff);
      Atom<bf16>::mma(W.w[ks], b, acc);
    }
    if (qok)
      store(q, oc0, pack4_bf16(fmaxf(acc[0] * scale + W.br[0], 0.f), fmaxf(acc[1] * scale + W.br[1], 0.f),
                               fmaxf(acc[2] * scale + W.br[2], 0.f), fmaxf(acc[3] * scale + W.br[3], 0.f)));
  }
}

constexpr int ACT_X_ELEMS = 84 * 84 * 4, ACT_A1_ELEMS = PatchGeom<LConv2Fwd>::SP, ACT_A2_ELEMS = PatchGeom<LConv3Fwd>::SP;
constexpr size_t ACT_ACT_BYTES = (size_t)(ACT_X_ELEMS + ACT_A1_ELEMS + ACT_A2_ELEMS) * 2; // stack + a1 + a2
constexpr size_t ACT_SMEM = 160 * 1024; // the rest of the LDS stages the layer weights (see act_conv_kernel)

// Weights: every wave needs its 16-channel atom of all three layers in registers (8 + 16 + 18 fragments), and 2-4
// waves share an atom, so loading straight from global memory moved 336 KB per workgroup - 7.7 of the kernel's 16 us
// (ablation).  Instead the workgroup copies each layer ONCE (152 KB in all, 19 coalesced 16-byte loads per thread issued
// at kernel start) and hands it to the waves through LDS regions that are free at that point: conv1's weights in the
// tail of the LDS, conv2's in the tail + the stack region (dead after conv1), conv3's over stack + a1 (dead after
// conv2).  Rows are padded by 16 B in LDS (conflict-free fragment reads).
template <int MODE> // 0: act on the stored stack; 1: ingest a given 84x84 frame first; 2: ingest a raw frame pair first
__global__ __launch_bounds__(512) void act_conv_kernel(ActConvParams P, ActIngestParams G) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16 *sx = reinterpret_cast<bf16 *>(smem), *s1 = sx + ACT_X_ELEMS, *s2 = s1 + ACT_A1_ELEMS;
  uint8_t *tail = smem + ACT_ACT_BYTES; // 60 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  using A1 = ActW<LConv1Full>;
  using A2 = ActW<LConv2FwdSmall>;
  using A3 = ActW<LConv3FwdSmall>;
  static_assert(32 * A1::PITCH <= ACT_SMEM - ACT_ACT_BYTES && 32 * A2::PITCH <= ACT_SMEM - ACT_ACT_BYTES &&
                    32 * A2::PITCH <= ACT_X_ELEMS * 2 && 64 * A3::PITCH <= (ACT_X_ELEMS + ACT_A1_ELEMS) * 2,
                "LDS regions for the weight images");
  A1 W1;
  A2 W2;
  A3 W3;
  u32x4 R[4];
  // packed u8 stack of sample n: 1764 16-byte vectors.  UNCONDITIONAL loads (index clamped): hipcc can then count
  // them and wait for these four alone while the weight loads issued behind them stay in flight.
  auto load_obs = [&](long n) {
    const long nn = n + P.map.n0;
    const long off = (nn / P.map.TP) * P.map.s1 + (nn % P.map.TP) * P.map.s0 + P.map.base;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(P.obs + off);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      R[i] = src[min(tid + 512 * i, 1763)];
  };
  // ---- fused ingest (MODE != 0).  Everything passes through LDS that is dead at this point: the raw frame pair
  // (67.2 KB, MODE 2) over the stack + a1 regions, the new frame's bytes and the palette table in the a2 region.
  // MODE 2 stages the pair with whole coalesced 16-byte loads (9 per thread, all in flight together - also the right
  // shape for frames that sit in mapped host memory) and gathers the 2-3 x 3 source pixels of an output pixel from LDS.
  uint8_t *sraw = smem;                                 // [2][210][160] raw pair (MODE 2)
  uint8_t *sfr = reinterpret_cast<uint8_t *>(s2);       // [7056] new-frame bytes
  uint8_t *slut = sfr + 7168;                           // [256]
  static_assert(2 * RAW_H * RAW_W <= (ACT_X_ELEMS + ACT_A1_ELEMS) * 2 && 7168 + 256 <= ACT_A2_ELEMS * 2, "ingest LDS map");
  auto ingest = [&](long n) {
    uint32_t start_byte = 0; // (requested first: over PCIe it lands while the frame is staged)
    if (G.start_bytes)
      start_byte = G.start_bytes[n];
    const long nn = n + P.map.n0;
    const long off = (nn / P.map.TP) * P.map.s1 + (nn % P.map.TP) * P.map.s0 + P.map.base; // acted slot, u32 pixels
    if constexpr (MODE == 1) {
      const u32x4 f = reinterpret_cast<const u32x4 *>(G.frames + n * (long)FRAME_PIX)[min(tid, 440)];
      if (tid < 441)
        reinterpret_cast<u32x4 *>(sfr)[tid] = f;
    } else {
      // The palette table is applied ONCE per raw byte while the pair is staged (67 K lookups instead of the 106 K an
      // output-pixel-wise lookup needs: source pixels are shared by neighbouring windows); an output pixel then reads
      // two aligned dwords per source row (ds_read2_b32), shifts its 2-3 bytes down and adds them with one v_sad_u8.
      constexpr int NRV = 2 * RAW_H * RAW_W / 16, NRL = (NRV + 511) / 512; // 4200 vectors, 9 per thread
      const u32x4 *raw = reinterpret_cast<const u32x4 *>(G.frames + n * (long)(2 * RAW_H * RAW_W));
      u32x4 Q[NRL];
#pragma unroll
      for (int i = 0; i < NRL; ++i)
        Q[i] = raw[min(tid + 512 * i, NRV - 1)];
      if (tid < 256)
        slut[tid] = G.lut[tid];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NRL; ++i) {
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uint32_t w = Q[i][d];
          o[d] = (uint32_t)slut[w & 255u] | ((uint32_t)slut[(w >> 8) & 255u] << 8) |
                 ((uint32_t)slut[(w >> 16) & 255u] << 16) | ((uint32_t)slut[w >> 24] << 24);
        }
        if (tid + 512 * i < NRV)
          reinterpret_cast<u32x4 *>(sraw)[tid + 512 * i] = o;
      }
      __syncthreads();
      const uint32_t *sraw32 = reinterpret_cast<const uint32_t *>(sraw);
      for (int pix = tid; pix < FRAME_PIX; pix += 512) {
        const int i = pix / 84, j = pix - i * 84;
        const int y0 = (i * RAW_H) / 84, x0 = (j * RAW_W) / 84, x1 = ((j + 1) * RAW_W + 83) / 84; // 3 rows, 2-3 cols
        const bool wide = (x1 - x0) == 3;
        const uint32_t keep = wide ? 0x00FFFFFFu : 0x0000FFFFu;
        int best = 0;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int a = f * (RAW_H * RAW_W) + y0 * RAW_W + x0; // byte offset of the window's first pixel
          const int sh = (a & 3) * 8;
          uint32_t sum = 0;
#pragma unroll
          for (int y = 0; y < 3; ++y) {
            const int ai = (a + y * RAW_W) >> 2; // (RAW_W % 4 == 0: the same byte phase in every row)
            const uint64_t two = (uint64_t)sraw32[ai] | ((uint64_t)sraw32[ai + 1] << 32);
            sum = __builtin_amdgcn_sad_u8((uint32_t)(two >> sh) & keep, 0u, sum);
          }
          best = max(best, (int)rintf((float)sum / (wide ? 9.0f : 6.0f))); // area mean in f32, round-half-even
        }
        sfr[pix] = (uint8_t)min(best, 255);
      }
    }
    { // the previous slot's stack: requested now, needed after the barrier
      const u32x4 *src = reinterpret_cast<const u32x4 *>(G.obs_rw + off + G.src_delta);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        R[i] = src[min(tid + 512 * i, 1763)];
    }
    __syncthreads();
    const bool st = G.start_bytes ? start_byte != 0 : ((G.sbits.w[n >> 5] >> (n & 31)) & 1u) != 0;
    u32x4 *dst = reinterpret_cast<u32x4 *>(G.obs_rw + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 512 * i;
      if (v < 1764) {
        const uint32_t fb = reinterpret_cast<const uint32_t *>(sfr)[v]; // the new frame's 4 pixels of this vector
        u32x4 nvw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t b = (fb >> (8 * j)) & 255u;
          nvw[j] = st ? b * 0x01010101u : ((R[i][j] << 8) | b); // rollout.cc:184-196, byte 0 = newest frame
        }
        R[i] = nvw;
        dst[v] = nvw; // Buffer::add's observation copy: the stack of the acted slot
      }
    }
    if constexpr (MODE == 2)
      __syncthreads(); // the widening step overwrites the staged raw pair: every wave must be done gathering from it
  };
  auto widen = [&]() {
    auto pk = [](uint32_t lo, uint32_t hi) { return pack_u8_pair_bf16(lo, hi); };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 512 * i;
      if (v < 1764) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const uint32_t w0 = R[i][2 * d], w1 = R[i][2 * d + 1];
          reinterpret_cast<u32x4 *>(sx)[2 * v + d] =
              u32x4{pk(w0 & 255u, (w0 >> 8) & 255u), pk((w0 >> 16) & 255u, w0 >> 24),
                    pk(w1 & 255u, (w1 >> 8) & 255u), pk((w1 >> 16) & 255u, w1 >> 24)};
        }
      }
    }
  };
  auto conv1 = [&]() {
    act_phase<LConv1Full>(sx, W1, 1.0f / 255.0f, wave, lane, [&](int q, int oc, u32x2 v) {
      *reinterpret_cast<u32x2 *>(s1 + (q / 20) * PatchGeom<LConv2Fwd>::RP + (q % 20) * LConv2Fwd::CP + oc) = v;
    });
  };
  auto conv2 = [&]() {
    act_phase<LConv2FwdSmall>(s1, W2, 1.0f, wave, lane, [&](int q, int oc, u32x2 v) {
      *reinterpret_cast<u32x2 *>(s2 + (q / 9) * PatchGeom<LConv3Fwd>::RP + (q % 9) * LConv3Fwd::CP + oc) = v;
    });
  };
  auto conv3 = [&](long n) {
    bf16 *out = P.a3 + n * (long)(49 * 64);
    act_phase<LConv3FwdSmall>(s2, W3, 1.0f, wave, lane,
                              [&](int q, int oc, u32x2 v) { *reinterpret_cast<u32x2 *>(out + q * 64 + oc) = v; });
  };
  // vector v of a layer's flat [rows][K] weight array -> its place in an LDS image with padded rows
  auto img_off = [](int v, int vpr, int pitch) { return (v / vpr) * pitch + (v % vpr) * 16; };

  long n = blockIdx.x;
  if (n >= P.ns)
    return;
  // ---- first sample: observation + ONE copy of every layer's weights per workgroup, all requested up front
  if constexpr (MODE == 0)
    load_obs(n);
  u32x4 S1[2], S2[8], S3[9];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    S1[i] = reinterpret_cast<const u32x4 *>(P.w1)[tid + 512 * i];
  // (with a fused ingest the conv2 / conv3 weights are requested after it - they would only sit in 68 registers while
  // the ingest needs them for its own loads - and land during the widening step and conv1)
  auto load_w23 = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      S2[i] = reinterpret_cast<const u32x4 *>(P.w2)[tid + 512 * i];
#pragma unroll
    for (int i = 0; i < 9; ++i)
      S3[i] = reinterpret_cast<const u32x4 *>(P.w3)[tid + 512 * i];
  };
  if constexpr (MODE == 0)
    load_w23();
  W1.load_bias(P.b1, wave, lane);
  W2.load_bias(P.b2, wave, lane);
  W3.load_bias(P.b3, wave, lane);
  if constexpr (MODE != 0) {
    ingest(n);
    load_w23();
  }
  widen();
#pragma unroll
  for (int i = 0; i < 2; ++i) // conv1 weights [32][256] -> tail
    *reinterpret_cast<u32x4 *>(tail + img_off(tid + 512 * i, 32, A1::PITCH)) = S1[i];
  __syncthreads();
  W1.load_lds(tail + (wave % A1::MA) * 16 * A1::PITCH, lane);
  conv1();
  __syncthreads(); // stack and conv1's weight image are dead
#pragma unroll
  for (int i = 0; i < 8; ++i) { // conv2 weights [64][512]: rows 0-31 -> tail, rows 32-63 -> stack region
    const int v = tid + 512 * i, row = v / 64;
    uint8_t *dst = row < 32 ? tail : smem - 32 * A2::PITCH;
    *reinterpret_cast<u32x4 *>(dst + img_off(v, 64, A2::PITCH)) = S2[i];
  }
  __syncthreads();
  {
    const int og = wave % A2::MA;
    W2.load_lds((og < 2 ? tail : smem - 32 * A2::PITCH) + og * 16 * A2::PITCH, lane);
  }
  conv2();
  __syncthreads(); // a1 and conv2's weight image are dead
#pragma unroll
  for (int i = 0; i < 9; ++i) // conv3 weights [64][576] -> stack + a1 region
    *reinterpret_cast<u32x4 *>(smem + img_off(tid + 512 * i, 72, A3::PITCH)) = S3[i];
  __syncthreads();
  W3.load_lds(smem + (wave % A3::MA) * 16 * A3::PITCH, lane);
  conv3(n);
  __syncthreads();
  // ---- further samples of this workgroup (more environments than CUs): the weights stay in registers
  for (n += gridDim.x; n < P.ns; n += gridDim.x) {
    if constexpr (MODE == 0)
      load_obs(n);
    else
      ingest(n);
    widen();
    __syncthreads();
    conv1();
    __syncthreads();
    conv2();
    __syncthreads();
    conv3(n);
    __syncthreads();
  }
}

// multi-sample variants for 4-wave workgroups (two or four independent workgroups per CU overlap one
// workgroup's staging / barrier with another's MFMAs)
struct LConv2FwdW4 : LConv2Fwd {
  static constexpr int SB = 1;
};

// conv3 fwd at training sizes: ONE 16-channel atom per wave (4 channel groups), 2 samples per group = exactly 7 atoms
// per wave, static atom loop; 4 waves, two workgroups per CU.  33.7 -> 30.1 us: with half the weights in registers
// the compiler keeps more LDS fragments in flight.  (The same for conv3 dgrad measured slower: 58 vs 45 us.)
struct LConv3Fwd1W4 : LConv3Fwd {
  static constexpr int SB = 2, OG = 4, ONE_ATOM = 1, STATIC_ATOMS = 1;
};
struct LConv2DgradW4 : LConv2Dgrad {
  static constexpr int SB = 1, PRELOAD_GATES = 1;
};
template <class L> constexpr size_t conv_patch_smem() {
  return (size_t)2 * ((L::SB * PatchGeom<L>::SP + 63) / 64 * 64) * 2;
}

// ================================================================================================
// wgrad: dW[oc][j] = sum over (sample, pixel) of dY[pixel][oc] * im2col(X)[pixel][j]
// ================================================================================================
// (conv1's weight gradient has its own kernel, conv1_wgrad.hpp; its implicit-GEMM instantiation of the kernel below - packed
// uint8 input, producer / consumer wave specialisation - was the round-1/2 form and is gone)
struct LConv2Wgrad {
  using InT = bf16;
  static constexpr int IN_ELEMS = 400 * 32, PIX = 81, OW = 9, S = 2, IW = 20, C = 32, KW = 4, OC = 64, NJ = 512, SB = 1,
                       MI = 4, NI = 4, WM = 1, GPS = 1, GSTRIDE = 0, CP = 40; // wave: all 4 oc atoms x 4 of the 32 j atoms
};
struct LConv3Wgrad {
  using InT = bf16;
  static constexpr int IN_ELEMS = 81 * 64, PIX = 49, OW = 7, S = 1, IW = 9, C = 64, KW = 3, OC = 64, NJ = 576, SB = 3,
                       MI = 2, NI = 9, WM = 2, GPS = 1, GSTRIDE = 0, CP = 80; // wave: 2 of 4 oc atoms x 9 of 36 j atoms
};

struct WgradParams {
  const void *x;    // layer input (bf16 NHWC, or packed u8 stacks)
  const bf16 *dy;   // [ns][PIX][OC]
  float *slab_w;    // [gridDim.x][OC][NJ]
  float *slab_b;    // [gridDim.x][OC]
  long ns;
  SampleMap map;
  float scale;
};

template <class L> __global__ __launch_bounds__(512) void conv_wgrad_patch_kernel(WgradParams P) {
  static_assert(sizeof(typename L::InT) == 2, "bf16 layer input (conv1 has its own kernel)");
  constexpr int PATCH = L::IN_ELEMS;
  constexpr int KPIX = L::SB * L::PIX;            // reduction length per group
  constexpr int KS = (KPIX + 31) / 32;            // atom-k steps per group (tail rows of dY are zero)
  constexpr int DYS = L::OC + 16;                 // dY tile row stride: (OC/2+8) dwords = 8 (mod 16)
  constexpr int LPATCH = PATCH / L::C * L::CP;    // padded pixel pitch (conflict-free transposed reads)
  constexpr int X_ELEMS = (L::SB * LPATCH + 63) / 64 * 64;
  constexpr int DY_ELEMS = KS * 32 * DYS;
  constexpr int BUF_ELEMS = X_ELEMS + DY_ELEMS;
  constexpr int NPROD = 512; // every thread stages
  constexpr int NI = L::NI;
  constexpr int XV = L::SB * PATCH * 2 / 16, NXV = (XV + NPROD - 1) / NPROD;
  constexpr int DV = KPIX * L::OC / 8, NDV = (DV + NPROD - 1) / NPROD; // dY source vectors (8 bf16)
  constexpr int VPR = L::OC / 8;                  // dY vectors per pixel row
  constexpr int SEG = L::KW * L::C;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16 *sbuf = reinterpret_cast<bf16 *>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % L::WM, wn = wave / L::WM;
  const int li = lane & 15, lg = lane >> 4;

  f32x4 acc[L::MI][NI];
#pragma unroll
  for (int i = 0; i < L::MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // this thread's 8 fixed channels (tid % VPR)

  // B-operand (im2col) element offsets of this lane: for tr-read r of k-step ks the lane supplies the
  // address of pixel (32ks + 16r + 4lg + (li>>2)), columns j0 + 4(li&3)..+3.  Pixel -> patch offset
  // depends only on (ks, r): precomputed once (same for every group).
  int pixoff[KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int q = ks * 32 + r * 16 + 4 * lg + (li >> 2);
      q = min(q, KPIX - 1); // tail pixels: dY rows are zero, any finite in-range patch data will do
      const int s = q / L::PIX, p = q - s * L::PIX, oy = p / L::OW, ox = p - oy * L::OW;
      pixoff[ks][r] = s * LPATCH + ((oy * L::S) * L::IW + ox * L::S) * L::CP;
    }
  int joff[NI]; // column part: j = 16*(atom) + 4(li&3) -> (kh, kw, c)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int jj = (wn * NI + j) * 16 + 4 * (li & 3);
    joff[j] = ((jj / SEG) * L::IW + (jj % SEG) / L::C) * L::CP + jj % L::C;
  }

  const long nunits = P.ns * L::GPS; // unit = sample or half sample; dY of unit u = rows [u*PIX, (u+1)*PIX)
  const long ngroups = (nunits + L::SB - 1) / L::SB;
  struct Regs {
    u32x4 x[NXV], d[NDV];
  };
  Regs R0; // one group ahead (a second register set spills here)
  // Loads are UNCONDITIONAL (indices clamped instead of predicated, groups past the end re-read the last one): only
  // then can hipcc count the outstanding loads and wait for ONE register set (vmcnt(N)) instead of draining both.
  auto gload = [&](Regs &R, long grp) {
    u32x4 (&RX)[NXV] = R.x;
    u32x4 (&RD)[NDV] = R.d;
    const long n0 = min(grp, ngroups - 1) * L::SB;
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int v = min(tid + NPROD * i, XV - 1);
      const long u = min(n0 + v / (PATCH / 8), nunits - 1); // unit of this vector (clamped: tail group)
      RX[i] = reinterpret_cast<const u32x4 *>(static_cast<const bf16 *>(P.x) + u * PATCH)[v % (PATCH / 8)];
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int v = min(tid + NPROD * i, DV - 1);
      const long u = min(n0 + (v / VPR) / L::PIX, nunits - 1);
      RD[i] = reinterpret_cast<const u32x4 *>(P.dy + u * (long)(L::PIX * L::OC))[v % (L::PIX * VPR)];
    }
  };
  // stages group `grp` from R; units past the end (tail group, prefetch beyond the last group) are staged as zeros
  auto swrite = [&](Regs &R, int buf, long grp) {
    u32x4 (&RX)[NXV] = R.x;
    u32x4 (&RD)[NDV] = R.d;
    bf16 *dx = sbuf + (size_t)buf * BUF_ELEMS, *dd = dx + X_ELEMS;
    const long u0 = grp * L::SB;
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int v = tid + NPROD * i;
      if (v < XV) {
        if (u0 + v / (XV / L::SB) >= nunits)
          RX[i] = zero16();
        constexpr int VPP = L::C / 8;
        *reinterpret_cast<u32x4 *>(dx + (v / VPP) * L::CP + (v % VPP) * 8) = RX[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int v = tid + NPROD * i;
      if (v < DV) {
        const int row = v / VPR, cv = v - row * VPR;
        if (u0 + row / L::PIX >= nunits)
          RD[i] = zero16();
        *reinterpret_cast<u32x4 *>(dd + row * DYS + cv * 8) = RD[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { // bias gradient: running column sums of what this thread stages
          bsum[2 * e] += bf16_bits_to_f32(RD[i][e] & 0xFFFFu);
          bsum[2 * e + 1] += bf16_bits_to_f32(RD[i][e] >> 16);
        }
      }
    }
  };
  // zero the dY tail rows [KPIX, KS*32) of both buffers once (never overwritten afterwards)
  for (int b = 0; b < 2; ++b)
    for (int e = tid; e < (KS * 32 - KPIX) * DYS; e += 512)
      sbuf[(size_t)b * BUF_ELEMS + X_ELEMS + KPIX * DYS + e] = (bf16)0.f;

  typedef __attribute__((address_space(3))) bf16x4 *lds4;
  auto multiply = [&](int buf) {
    const bf16 *px = sbuf + (size_t)buf * BUF_ELEMS, *pd = px + X_ELEMS;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 fa[L::MI], fb[NI];
#pragma unroll
      for (int i = 0; i < L::MI; ++i)
        fa[i] = KFrag<bf16>::read(pd, DYS, ks * 32, (wm * L::MI + i) * 16, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const u32x2 lo = __builtin_bit_cast(
            u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(px + pixoff[ks][0] + joff[j])));
        const u32x2 hi = __builtin_bit_cast(
            u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(px + pixoff[ks][1] + joff[j])));
        fb[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
      for (int i = 0; i < L::MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          Atom<bf16>::mma(fa[i], fb[j], acc[i][j]);
    }
  };
  // iteration i multiplies LDS buffer i & 1 (group g_i), then stages group g_{i+1} and requests group g_{i+2}
  const long gs = gridDim.x;
  long grp = blockIdx.x;
  gload(R0, grp);
  swrite(R0, 0, grp);
  gload(R0, grp + gs);
  __syncthreads();
  for (int it = 0; grp < ngroups; grp += gs, ++it) {
    multiply(it & 1);
    swrite(R0, (it + 1) & 1, grp + gs);
    gload(R0, grp + 2 * gs);
    __syncthreads();
  }
  // ---- one slab per workgroup
  float *ow = P.slab_w + (long)blockIdx.x * L::OC * L::NJ;
#pragma unroll
  for (int i = 0; i < L::MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = (wm * L::MI + i) * 16 + lg * 4 + r;
#pragma unroll
      for (int j = 0; j < NI; ++j)
        ow[(long)m * L::NJ + (wn * NI + j) * 16 + li] = acc[i][j][r] * P.scale;
    }
  // bias: threads with equal tid % VPR hold the same 8 channels; ordered LDS reduction (deterministic)
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    red[tid * 8 + e] = bsum[e];
  __syncthreads();
  if (tid < L::OC) {
    const int cv = tid / 8, e = tid % 8;
    float s = 0.f;
    for (int t = cv; t < 512; t += VPR)
      s += red[t * 8 + e];
    P.slab_b[(long)blockIdx.x * L::OC + tid] = s;
  }
}

template <class L> constexpr size_t conv_wgrad_patch_smem() {
  constexpr int KPIX = L::SB * L::PIX, KS = (KPIX + 31) / 32;
  return (size_t)2 * (((L::SB * (L::IN_ELEMS / L::C * L::CP) + 63) / 64 * 64) + KS * 32 * (L::OC + 16)) * 2;
}

} // namespace aleppo
