// conv_bwd_fused.hpp - the tail of the update's backward pass in ONE launch (bf16):
//
//   conv2 data gradient    dz1[y][x][c]  = (a1[y][x][c] > 0) * sum_{kh,kw,oc} W2[oc][kh][kw][c] * dz2[(y-kh)/2][(x-kw)/2][oc]
//   conv2 weight gradient  dW2[oc][kh][kw][c] += sum_{oy,ox} dz2[oy][ox][oc] * a1[2oy+kh][2ox+kw][c],   db2 += sum dz2
//   conv1 weight gradient  dW1[oc][kh][kw][c] += sum_{oy,ox} dz1[oy][ox][oc] * obs[4oy+kh][4ox+kw][c],  db1 += sum dz1
//
// As three launches (conv_patch_kernel<LConv2DgradW4>, conv_wgrad_patch_kernel<LConv2Wgrad>, conv1_wgrad_shift_kernel) they
// move 252 + 147 + 220 MB per 4096-sample minibatch: dz2 and a1 are read twice and dz1 (105 MB) is written and read back.
// Here a workgroup (4 waves = one per SIMD, the whole register file each, persistent over samples - conv_fwd_fused.hpp's
// scheme) reads dz2, a1 and the observations of a sample ONCE (262 MB per minibatch) and dz1 never leaves the CU:
//   P0  stage dz2 (zero-bordered 11 x 11 image for the dgrad + pixel-major tile for the wgrad) and a1 into LDS
//   P1  conv2 dgrad: wave = one parity class of the 20 x 20 grid (2 x 2 live taps of the 4 x 4 stride-2 kernel), W2d resident
//       in registers; ReLU gate from the a1 image; the result goes, as bf16, into the tap-shift tile of conv1's weight gradient
//   P2  conv2 wgrad: wave = one kernel row kh; 32 accumulator tiles (4 channel atoms x 4 kw x 2 channel halves) in VGPRs
//   P3  widen the sample's packed uint8 stack into LDS (over the dead dz2 / a1 images)
//   P4  conv1 wgrad, tap-shift form of conv1_wgrad.hpp on the whole 21 x 21 cell grid: the four waves split the k-steps;
//       34 accumulator tiles per wave in AGPRs, summed across the waves once at the end (fixed order)
// Every LDS read and MFMA of P1 / P2 / P4 is spelled out (inline assembly, hand-counted waits: the rules are in
// conv_fwd_fused.hpp); the next sample's tensors are requested into AGPRs a whole sample ahead (no stores in the loop: the
// `vmcnt` counts are the other stream of prefetch loads).  One fp32 slab of dW2 / db2 / dW1 / db1 per workgroup, as before.
#pragma once
#include "conv_fwd_fused.hpp"

namespace aleppo {

struct ConvBwdParams {
  const bf16 *dz2, *a1, *w2d; // [ns][81][64], [ns][400][32], [4 classes][32 c][(a,b,oc) 256]
  const uint32_t *obs;        // packed stacks, located through map
  SampleMap map;
  float *sw2, *sb2, *sw1, *sb1; // [grid][64][512], [grid][64], [grid][32][256], [grid][32]
  long ns;
  float scale1; // 1/255 (folded into dW1)
};

namespace cb {
constexpr int NT = 256;
// region A, phases 0-2: I1 = dz2 with a zero border (dgrad B operand, ds_read_b128).  Pixel pitch 80 elements, row pitch 928:
// 4.0 LDS cycles per fragment read in the simulation of the gfx950 lane groups (tests/tools/lds_conflicts.py's model; the
// first choice, 72 / 848, looked conflict-free for 16 consecutive lanes and costs 7.4 - the groups are not consecutive lanes)
constexpr int CP1 = 80, PR1 = 928, I1_ELEMS = 11 * PR1;
// I2 = dz2 pixel-major (wgrad A operand, transposed reads): 96 rows (81 pixels + zero rows) x (64 + 16)
constexpr int T2S = 80, I2_ELEMS = 96 * T2S;
// I3 = a1 (wgrad B operand, transposed reads; ReLU gates): 400 pixels x (32 + 8)
constexpr int XP = 40, I3_ELEMS = 400 * XP;
constexpr int A_ELEMS = I1_ELEMS + I2_ELEMS + I3_ELEMS; // 33,008 bf16
// region A, phases 3-4: X = the widened stack [84][84][4]
constexpr int XROW = 84 * 4, XF_ELEMS = 84 * XROW;
static_assert(XF_ELEMS <= A_ELEMS, "the widened stack lies over the dz2 / a1 images");
// region B: conv1 wgrad's dY tile on the 21-wide cell grid (conv1_wgrad.hpp), whole frame: 16 k-steps of 32 cells (14 hold
// cells; every wave runs 4), QOFF zero rows in front for the shifted views
constexpr int GW = 21, QOFF = 32, KS1 = 16, DYS = 48, DYROWS = QOFF + KS1 * 32, DY_ELEMS = DYROWS * DYS;
constexpr int KMAX = 21 * 21 - 1;
constexpr int NT1 = 4 * 2 * 4 + 2; // conv1 wgrad accumulator tiles per wave (+ 2 bias tiles)
constexpr size_t SMEM_LOOP = (size_t)(A_ELEMS + DY_ELEMS) * 2, SMEM_RED = (size_t)4 * NT1 * 64 * 16;
constexpr size_t SMEM = SMEM_LOOP > SMEM_RED ? SMEM_LOOP : SMEM_RED;
constexpr int DV = 81 * 8, AV = 400 * 4, OV = 1764; // 16-byte source vectors per sample: dz2, a1, stack
constexpr int NZ = 40 * 8 + 15 * 10;               // vectors re-zeroed per sample (I1's border: 64 channels, I2's rows 81..95)
constexpr int dgrad_koff(int ks) {                 // byte offset of k-step ks from the class-grid cell (Y, X)
  const int tap = ks / 2, a = tap >> 1, b = tap & 1;
  return 2 * ((1 - a) * PR1 + (1 - b) * CP1 + (ks % 2) * 32);
}
// Schedule of the "group of 8 items" pipelines (P2, P4): before the loop G(0) [8 reads], I(0..2) [2 reads each]; step t:
// { t % 8 == 4: G(t / 8 + 1) [8] } { I(t + 3) [2] } WAIT(t) MFMAs.  WAIT(t) = lgkmcnt(number of reads issued behind the
// youngest read step t needs): I(t), and for the first item of a group also G(group).
constexpr int g8_wait(int t, int NG) {
  int issued = 8 + 6, last_i = 0, last_g = 8; // after the prologue: G(0) ends at 8, I(0) at 10, I(1) 12, I(2) 14
  int item_end[40] = {0}, grp_end[8] = {0};
  grp_end[0] = 8;
  item_end[0] = 10, item_end[1] = 12, item_end[2] = 14;
  for (int s = 0; s <= t; ++s) {
    if (s % 8 == 4 && s / 8 + 1 < NG) {
      issued += 8;
      grp_end[s / 8 + 1] = issued;
    }
    if (s + 3 < 8 * NG) {
      issued += 2;
      item_end[s + 3] = issued;
    }
  }
  last_i = item_end[t];
  last_g = (t % 8 == 0) ? grp_end[t / 8] : 0;
  const int need = last_i > last_g ? last_i : last_g;
  return issued - need;
}
static_assert(g8_wait(0, 3) == 6 && g8_wait(4, 3) == 14 && g8_wait(7, 3) == 6 && g8_wait(8, 3) == 6 && g8_wait(23, 3) == 0,
              "pipeline wait counts");
} // namespace cb

__device__ __forceinline__ void ff_write(uint32_t addr, const u32x4 &v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ff_pin(u32x4 &v) { asm volatile("" : "+v"(v)); }
template <int OFF> __device__ __forceinline__ void ff_read_tr(u32x2 &dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// acc += A x B with the 128-bit operands assembled from two transposed 64-bit reads each; accumulator in a VGPR / AGPR tuple
__device__ __forceinline__ void ff_mfma_v(f32x4 &acc, const u32x2 &alo, const u32x2 &ahi, const u32x2 &blo, const u32x2 &bhi) {
  const u32x4 a = {alo[0], alo[1], ahi[0], ahi[1]}, b = {blo[0], blo[1], bhi[0], bhi[1]};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void ff_mfma_a(f32x4 &acc, const u32x2 &alo, const u32x2 &ahi, const u32x2 &blo, const u32x2 &bhi) {
  const u32x4 a = {alo[0], alo[1], ahi[0], ahi[1]}, b = {blo[0], blo[1], bhi[0], bhi[1]};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void ff_mfma_a1(f32x4 &acc, const u32x2 &alo, const u32x2 &ahi, const u32x4 &b) {
  const u32x4 a = {alo[0], alo[1], ahi[0], ahi[1]};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// the counted wait of a group-of-8 pipeline step; it names the registers it covers (their 64-bit halves are merged into
// 128-bit operands by ordinary code, which must not move in front of the wait)
template <int N> __device__ __forceinline__ void ff_wait_item(u32x2 &a, u32x2 &b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void ff_wait_item_group(u32x2 &a, u32x2 &b, u32x2 (&g)[4][2]) {
  asm volatile("s_waitcnt lgkmcnt(%10)"
               : "+v"(a), "+v"(b), "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[1][0]), "+v"(g[1][1]), "+v"(g[2][0]), "+v"(g[2][1]),
                 "+v"(g[3][0]), "+v"(g[3][1])
               : "n"(N));
}
template <int NVM, int NR> struct FFWaitVm;
template <int NVM> struct FFWaitVm<NVM, 10> { // vmcnt(NVM) for 10 prefetch registers (AGPRs written by asm loads)
  static __device__ __forceinline__ void wait(u32x4 (&r)[10]) {
    asm volatile("s_waitcnt vmcnt(%10)"
                 : "+a"(r[0]), "+a"(r[1]), "+a"(r[2]), "+a"(r[3]), "+a"(r[4]), "+a"(r[5]), "+a"(r[6]), "+a"(r[7]), "+a"(r[8]),
                   "+a"(r[9])
                 : "n"(NVM)
                 : "memory");
  }
};
template <int NVM> struct FFWaitVm<NVM, 7> {
  static __device__ __forceinline__ void wait(u32x4 (&r)[7]) {
    asm volatile("s_waitcnt vmcnt(%7)"
                 : "+a"(r[0]), "+a"(r[1]), "+a"(r[2]), "+a"(r[3]), "+a"(r[4]), "+a"(r[5]), "+a"(r[6])
                 : "n"(NVM)
                 : "memory");
  }
};

// ABL: timing-only ablations (wrong results): 1 no conv2 dgrad, 2 no conv2 wgrad, 4 no conv1 wgrad, 8 no staging stores, 16 no
// widening
template <int ABL>
__global__ __launch_bounds__(cb::NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_bwd_fused_kernel(ConvBwdParams P) {
  using namespace cb;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  typedef __attribute__((address_space(3))) uint8_t *lds_ptr;
  bf16 *sA = reinterpret_cast<bf16 *>(smem), *sDY = sA + A_ELEMS;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4, li = fr, lg = fg;
  const long gs = gridDim.x;
  long n = blockIdx.x;
  if (n >= P.ns)
    return;

  // ---- conv2 dgrad weights of this wave's parity class: rows (fr >> 2) * 8 + a * 4 + (fr & 3) -> 8 consecutive channels per lane
  u32x4 Wd[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      Wd[a][ks] = *reinterpret_cast<const u32x4 *>(P.w2d + (long)(wave * 32 + (fr >> 2) * 8 + a * 4 + (fr & 3)) * 256 + ks * 32 +
                                                   fg * 8);
  // (hipcc must see these loads land BEFORE the loop: left pending, its wait-count pass - which does not know the prefetch
  // loads issued by inline assembly - waits for them inside the loop with counts that drain the prefetch instead)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    asm volatile("" : "+a"(Wd[0][ks]));
    asm volatile("" : "+v"(Wd[1][ks]));
  }
  // ---- accumulators: conv2 wgrad tile (i, jl) = channel atom i x (kw = jl >> 1, input channels 16 (jl & 1) .. + 15) of kernel row
  // kh = wave; conv1 wgrad tile (sh, i, j) = tap block (a, b) = (sh >> 1, sh & 1) x channel atom i x cell row j (k-split: summed
  // over the waves at the end), + the bias tiles
  f32x4 acc2[4][8], acc1[4][2][4], accb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sh = 0; sh < 4; ++sh)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc1[sh][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // db2: this thread's 8 channels (tid & 7)

  // ---- the dY tile is zeroed once: the rows the dgrad writes are the same for every sample, the others stay zero
  for (int e = tid; e < DY_ELEMS / 8; e += NT)
    reinterpret_cast<u32x4 *>(sDY)[e] = zero16();

  // ---- prefetch registers (AGPRs): RA = 3 dz2 + 7 a1 vectors, RX = 7 stack vectors per thread and sample
  u32x4 RA[10], RX[7];
  auto prefetch_a = [&](long m) { // unconditional, clamped
    const long mm = min(m, P.ns - 1);
    const u32x4 *pd = reinterpret_cast<const u32x4 *>(P.dz2 + mm * (81 * 64));
    const u32x4 *pa = reinterpret_cast<const u32x4 *>(P.a1 + mm * (400 * 32));
#pragma unroll
    for (int i = 0; i < 3; ++i)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(RA[i]) : "v"(pd + min(tid + NT * i, DV - 1)) : "memory");
#pragma unroll
    for (int i = 0; i < 7; ++i)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(RA[3 + i]) : "v"(pa + min(tid + NT * i, AV - 1)) : "memory");
  };
  auto prefetch_x = [&](long m) {
    const long nn = min(m, P.ns - 1) + P.map.n0;
    const long off = (nn / P.map.TP) * P.map.s1 + (nn % P.map.TP) * P.map.s0 + P.map.base;
    const u32x4 *src = reinterpret_cast<const u32x4 *>(P.obs + off);
#pragma unroll
    for (int i = 0; i < 7; ++i)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(RX[i]) : "v"(src + min(tid + NT * i, OV - 1)) : "memory");
  };
  // element offsets (region A) of the vectors this thread re-zeroes per sample: I1's border pixels, I2's rows 81..95
  int zoff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = min(tid + NT * k, NZ - 1);
    if (idx < 320) {
      const int b = idx >> 3, part = idx & 7;
      const int y = b < 11 ? 0 : b < 22 ? 10 : 1 + (b - 22) / 2, x = b < 11 ? b : b < 22 ? b - 11 : ((b - 22) & 1) * 10;
      zoff[k] = y * PR1 + x * CP1 + part * 8;
    } else {
      const int j = idx - 320, row = 81 + j / 10, part = j - (j / 10) * 10;
      zoff[k] = I1_ELEMS + row * T2S + part * 8;
    }
  }

  // ---- lane constants of the wgrad phases
  const int lrow = 4 * lg + (li >> 2), lcol = 4 * (li & 3); // row / first column a lane supplies to a transposed read

  __syncthreads(); // dY tile zeroed
  prefetch_a(n);
  prefetch_x(n);
  auto pk = [](uint32_t lo, uint32_t hi) { return pack_u8_pair_bf16(lo, hi); };

  for (; n < P.ns; n += gs) {
    // =========================================================== P0: dz2 -> I1 + I2, a1 -> I3
    FFWaitVm<7, 10>::wait(RA); // behind these loads: the 7 stack loads
    if constexpr (!(ABL & 8)) {
      int zs; // (opaque zero: the ~20 staging addresses are recomputed per sample instead of living in registers)
      asm volatile("v_mov_b32 %0, 0" : "=v"(zs));
      const int tidz = tid + zs;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int v = tidz + NT * i;
        if (v < DV) {
          const int p = v >> 3, part = v & 7, oy = p / 9, ox = p - oy * 9;
          *reinterpret_cast<u32x4 *>(sA + (oy + 1) * PR1 + (ox + 1) * CP1 + part * 8) = RA[i];
          *reinterpret_cast<u32x4 *>(sA + I1_ELEMS + p * T2S + part * 8) = RA[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) { // db2
            bsum[2 * e] += bf16_bits_to_f32(RA[i][e] & 0xFFFFu);
            bsum[2 * e + 1] += bf16_bits_to_f32(RA[i][e] >> 16);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int v = tidz + NT * i;
        if (v < AV)
          *reinterpret_cast<u32x4 *>(sA + I1_ELEMS + I2_ELEMS + (v >> 2) * XP + (v & 3) * 8) = RA[3 + i];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (tid + NT * k < NZ)
          *reinterpret_cast<u32x4 *>(sA + zoff[k]) = zero16();
    }
    prefetch_a(n + gs);
    __syncthreads();
    // =========================================================== P1: dz1 of sample n -> dY tile
    if constexpr (!(ABL & 1)) {
      constexpr int APW = 7, KS = 8, N = APW * KS, D = 6;
      int z0; // (a zero hipcc cannot see through: the lane addresses below are recomputed per atom, not kept in registers)
      asm volatile("v_mov_b32 %0, 0" : "=v"(z0));
      const int py = wave >> 1, px = wave & 1, frz = fr + z0;
      uint32_t base = 0; // cell (Y, X) of the atom whose fragments are being requested
      // cell (Y, X) of atom ak's pixel on the 10 x 10 class grid; 24-bit multiplies (full rate; `*` compiles to the quarter-rate
      // v_mul_lo_u32, and with ~75 vector instructions per atom this phase was bound by them, not by its 16 MFMAs)
      auto atom_yx = [&](int ak, int &Y, int &X) {
        const int q = min(ak * 16 + frz, 99);
        Y = __mul24(q, 205) >> 11; // q / 10 for q < 1029
        X = q - __mul24(Y, 10);
      };
      const uint32_t c_base = lds0 + 16u * (uint32_t)fg;
      const uint32_t c_gate = lds0 + 2u * (uint32_t)(I1_ELEMS + I2_ELEMS + (py * 20 + px) * XP) + 16u * (uint32_t)fg;
      const uint32_t c_dy = lds0 + 2u * (uint32_t)(A_ELEMS + (QOFF + py * GW + px) * DYS) + 16u * (uint32_t)fg;
      u32x4 frag[D], gate[3];
      f32x4 dacc[2][2];
      uint32_t packed[4];
      auto issue = [&](auto I) {
        constexpr int i = decltype(I)::value, ak = i / KS, ks = i % KS;
        if constexpr (ks == 0) { // a new atom: its cell address, and its ReLU gate (read long before the epilogue needs it)
          int Y, X;
          atom_yx(ak, Y, X);
          base = c_base + (uint32_t)(__mul24(Y, 2 * PR1) + __mul24(X, 2 * CP1));
          ff_read<0>(gate[ak % 3], c_gate + (uint32_t)(__mul24(Y, 2 * 20 * XP * 2) + __mul24(X, 2 * XP * 2))); // pixel (2Y+py, 2X+px)
        }
        ff_read<dgrad_koff(ks)>(frag[i % D], base);
      };
      auto epilogue = [&](auto AK, auto PIECE) {
        constexpr int ak = decltype(AK)::value, piece = decltype(PIECE)::value, set = ak & 1;
        if constexpr (piece < 4) { // channels 2 piece, 2 piece + 1, gated by the stored bf16 activation
          const f32x4 &a = dacc[set][piece >> 1];
          constexpr int r = (piece & 1) * 2;
          // a1 > 0 as float <=> its bf16 bits > 0 as int16 (a1 is a ReLU output: +0, -0 or positive): per-half mask, 3 packed ops
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          const uint32_t gw = gate[ak % 3][piece]; // (by value first: bit-casting the vector element itself read element 0)
          const s16x2 g = __builtin_bit_cast(s16x2, gw);
          const s16x2 m = -__builtin_elementwise_min(__builtin_elementwise_max(g, s16x2{0, 0}), s16x2{1, 1});
          packed[piece] = pack2_bf16(a[r], a[r + 1]) & __builtin_bit_cast(uint32_t, m);
        } else {
          int Y, X;
          atom_yx(ak, Y, X);
          const u32x4 v = {packed[0], packed[1], packed[2], packed[3]};
          ff_write(c_dy + (uint32_t)(__mul24(Y, 2 * GW * DYS * 2) + __mul24(X, 2 * DYS * 2)), v); // grid row QOFF + 21 oy + ox
        }
      };
      static_for<D - 1>(issue);
      static_for<N>([&](auto I) {
        constexpr int i = decltype(I)::value, ak = i / KS, ks = i % KS, set = ak & 1;
        if constexpr (i + D - 1 < N)
          issue(IC<i + D - 1>{});
        ff_wait_lgkm<(N - 1 - i < D - 1 ? N - 1 - i : D - 1)>(); // (gate reads / dY stores in the queue only lengthen it)
        if constexpr (ks == 0) {
          ff_mfma0<true>(dacc[set][0], Wd[0][0], frag[i % D]);
          ff_mfma0<false>(dacc[set][1], Wd[1][0], frag[i % D]);
        } else {
          ff_mfma<true>(dacc[set][0], Wd[0][ks], frag[i % D]);
          ff_mfma<false>(dacc[set][1], Wd[1][ks], frag[i % D]);
        }
        if constexpr (ak > 0 && ks >= 1 && ks <= 6) {
          if constexpr (ks == 1) {
            ff_release(dacc[set ^ 1][0], dacc[set ^ 1][1]);
            ff_pin(gate[(ak - 1) % 3]); // landed: >= 6 younger reads have been waited for
          } else
            epilogue(IC<ak - 1>{}, IC<ks - 2>{});
        }
      });
      ff_release_nops(dacc[(APW - 1) & 1][0], dacc[(APW - 1) & 1][1]);
      ff_pin(gate[(APW - 1) % 3]);
      static_for<5>([&](auto PIECE) { epilogue(IC<APW - 1>{}, PIECE); });
    }
    // =========================================================== P2: dW2 += sample n (reads I2, I3 only: no barrier in front)
    if constexpr (!(ABL & 2)) {
      constexpr int NG = 3, NI = 8 * NG, DB = 4;
      u32x2 fa[2][4][2], fb[DB][2];
      // transposed read h of k-step ks supplies pixel row 32 ks + 16 h + lrow of dz2 (rows past 80: zero rows of I2, a1 address
      // clamped), columns lcol .. + 3 of the atom.  (Recomputed per sample from an opaque zero: kept across the phases these
      // addresses would cost registers the accumulators need.)
      int z2;
      asm volatile("v_mov_b32 %0, 0" : "=v"(z2));
      const int lrow2 = lrow + z2;
      const uint32_t a2base = lds0 + 2u * (uint32_t)(I1_ELEMS + lrow2 * T2S + lcol);
      uint32_t x2base[3][2];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int q = min(ks * 32 + h * 16 + lrow2, 80), oy = q / 9, ox = q - oy * 9;
          x2base[ks][h] = lds0 + 2u * (uint32_t)(I1_ELEMS + I2_ELEMS + ((2 * oy + wave) * 20 + 2 * ox) * XP + lcol);
        }
      auto issue_group = [&](auto G) { // dz2 tile fragments of k-step G: 4 channel atoms x 2 transposed reads
        constexpr int g = decltype(G)::value;
        static_for<4>([&](auto II) {
          constexpr int i = decltype(II)::value;
          ff_read_tr<2 * (g * 32 * T2S + i * 16)>(fa[g & 1][i][0], a2base);
          ff_read_tr<2 * ((g * 32 + 16) * T2S + i * 16)>(fa[g & 1][i][1], a2base);
        });
      };
      auto issue_item = [&](auto T) { // a1 fragment (kw, channel half) of k-step T / 8
        constexpr int t = decltype(T)::value, g = t / 8, jl = t % 8;
        constexpr int joff = 2 * ((jl >> 1) * XP + (jl & 1) * 16);
        ff_read_tr<joff>(fb[t % DB][0], x2base[g][0]);
        ff_read_tr<joff>(fb[t % DB][1], x2base[g][1]);
      };
      issue_group(IC<0>{});
      static_for<DB - 1>(issue_item);
      static_for<NI>([&](auto T) {
        constexpr int t = decltype(T)::value, g = t / 8, jl = t % 8;
        if constexpr (jl == 4 && g + 1 < NG)
          issue_group(IC<g + 1>{});
        if constexpr (t + DB - 1 < NI)
          issue_item(IC<t + DB - 1>{});
        if constexpr (jl == 0)
          ff_wait_item_group<g8_wait(t, NG)>(fb[t % DB][0], fb[t % DB][1], fa[g & 1]);
        else
          ff_wait_item<g8_wait(t, NG)>(fb[t % DB][0], fb[t % DB][1]);
        static_for<4>([&](auto II) {
          constexpr int i = decltype(II)::value;
          if constexpr (jl == 7) // (these four tiles live in AGPRs: the VGPR file is full)
            ff_mfma_a(acc2[i][jl], fa[g & 1][i][0], fa[g & 1][i][1], fb[t % DB][0], fb[t % DB][1]);
          else
            ff_mfma_v(acc2[i][jl], fa[g & 1][i][0], fa[g & 1][i][1], fb[t % DB][0], fb[t % DB][1]);
        });
      });
    }
    __syncthreads(); // dz2 / a1 images dead, dY tile complete
    // =========================================================== P3: widen the stack over region A
    FFWaitVm<10, 7>::wait(RX); // behind these loads: the 10 dz2 / a1 loads of the next sample
    int zw;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zw));
    if constexpr (!(ABL & 16))
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int v = tid + zw + NT * i;
      if (v < OV) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const uint32_t w0 = RX[i][2 * d], w1 = RX[i][2 * d + 1];
          reinterpret_cast<u32x4 *>(sA)[2 * v + d] = u32x4{pk(w0 & 255u, (w0 >> 8) & 255u), pk((w0 >> 16) & 255u, w0 >> 24),
                                                          pk(w1 & 255u, (w1 >> 8) & 255u), pk((w1 >> 16) & 255u, w1 >> 24)};
        }
      }
    }
    prefetch_x(n + gs);
    __syncthreads();
    // =========================================================== P4: dW1 += sample n (this wave's 4 of the 16 k-steps)
    if constexpr (!(ABL & 4)) {
      constexpr int NG = 4, NI = 8 * NG, DB = 4;
      u32x2 fx[2][4][2], fd[DB][2];
      // wave cw owns k-steps cw, cw + 4, cw + 8, cw + 12 of the 16 (conv1_wgrad.hpp's operand maps on the whole-frame grid)
      int z4;
      asm volatile("v_mov_b32 %0, 0" : "=v"(z4));
      const int lrow4 = lrow + z4;
      uint32_t x1base[4][2];
#pragma unroll
      for (int tl = 0; tl < 4; ++tl)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k = min(32 * (wave + 4 * tl) + 16 * h + lrow4, KMAX); // (cells past the frame meet zero dY rows)
          const int gy = k / GW, gx = k - gy * GW;
          x1base[tl][h] = lds0 + 2u * (uint32_t)((4 * gy * 84 + 4 * gx) * 4 + lcol);
        }
      const uint32_t a1base = lds0 + 2u * (uint32_t)(A_ELEMS + (32 * wave + lrow4) * DYS + lcol);
      const uint32_t one2 = 0x3F803F80u + (uint32_t)z4; // bf16 {1, 1} (built here: not a register held across the phases)
      const u32x4 ones = {one2, one2, one2, one2};
      auto issue_group = [&](auto G) { // cell rows 0..3 of k-step G's 32 cells: 4 x 2 transposed reads
        constexpr int g = decltype(G)::value;
        static_for<4>([&](auto JJ) {
          constexpr int j = decltype(JJ)::value;
          ff_read_tr<2 * (j * XROW)>(fx[g & 1][j][0], x1base[g][0]);
          ff_read_tr<2 * (j * XROW)>(fx[g & 1][j][1], x1base[g][1]);
        });
      };
      auto issue_item = [&](auto T) { // dY fragment: view shifted by (a, b) = (sh >> 1, sh & 1), channel atom i
        constexpr int t = decltype(T)::value, g = t / 8, u = t % 8, sh = u >> 1, i = u & 1;
        constexpr int row = QOFF + 128 * g - (sh >> 1) * GW - (sh & 1);
        ff_read_tr<2 * (row * DYS + 16 * i)>(fd[t % DB][0], a1base);
        ff_read_tr<2 * ((row + 16) * DYS + 16 * i)>(fd[t % DB][1], a1base);
      };
      issue_group(IC<0>{});
      static_for<DB - 1>(issue_item);
      static_for<NI>([&](auto T) {
        constexpr int t = decltype(T)::value, g = t / 8, u = t % 8, sh = u >> 1, i = u & 1;
        if constexpr (u == 4 && g + 1 < NG)
          issue_group(IC<g + 1>{});
        if constexpr (t + DB - 1 < NI)
          issue_item(IC<t + DB - 1>{});
        if constexpr (u == 0)
          ff_wait_item_group<g8_wait(t, NG)>(fd[t % DB][0], fd[t % DB][1], fx[g & 1]);
        else
          ff_wait_item<g8_wait(t, NG)>(fd[t % DB][0], fd[t % DB][1]);
        static_for<4>([&](auto JJ) {
          constexpr int j = decltype(JJ)::value;
          ff_mfma_a(acc1[sh][i][j], fd[t % DB][0], fd[t % DB][1], fx[g & 1][j][0], fx[g & 1][j][1]);
        });
        if constexpr (sh == 0)
          ff_mfma_a1(accb[i], fd[t % DB][0], fd[t % DB][1], ones); // bias gradient: column sums of the unshifted view
      });
    }
    __syncthreads(); // X and the dY tile are rewritten by the next sample
  }

  // ---- slabs.  (The last MFMAs' results are read by ordinary code below: explicit wait states first.)
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
  {
    float *ow = P.sw2 + (long)blockIdx.x * 64 * 512;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jl = 0; jl < 8; ++jl) {
        f32x4 v = acc2[i][jl];
        if (jl == 7)
          asm volatile("" : "+a"(v));
        else
          asm volatile("" : "+v"(v));
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ow[(long)(i * 16 + lg * 4 + r) * 512 + (wave * 4 + (jl >> 1)) * 32 + (jl & 1) * 16 + li] = v[r];
      }
  }
  // conv1: the four k-split parts are added in wave order through LDS
  f32x4 *red = reinterpret_cast<f32x4 *>(smem);
#pragma unroll
  for (int sh = 0; sh < 4; ++sh)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v = acc1[sh][i][j];
        asm volatile("" : "+a"(v));
        red[(wave * NT1 + (sh * 2 + i) * 4 + j) * 64 + lane] = v;
      }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f32x4 v = accb[i];
    asm volatile("" : "+a"(v));
    red[(wave * NT1 + 32 + i) * 64 + lane] = v;
  }
  __syncthreads();
  float *ow1 = P.sw1 + (long)blockIdx.x * 32 * 256, *ob1 = P.sb1 + (long)blockIdx.x * 32;
  for (int t = wave; t < NT1; t += 4) {
    f32x4 v = red[t * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 u = red[(w * NT1 + t) * 64 + lane];
      v = f32x4{v[0] + u[0], v[1] + u[1], v[2] + u[2], v[3] + u[3]};
    }
    if (t < 32) { // tile ((a,b), channel atom i, cell row r): columns (kh = 4a + r, kw = 4b + s, c), li = 4 s + c
      const int sh = t >> 3, i = (t >> 2) & 1, r = t & 3;
      const int col = ((4 * (sh >> 1) + r) * 8 + 4 * (sh & 1)) * 4 + li;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ow1[(long)(i * 16 + lg * 4 + e) * 256 + col] = v[e] * P.scale1;
    } else if (li == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ob1[(t - 32) * 16 + lg * 4 + e] = v[e];
    }
  }
  // db2: threads with equal tid & 7 hold the same 8 channels; ordered LDS reduction (deterministic)
  __syncthreads();
  float *redf = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    redf[tid * 8 + e] = bsum[e];
  __syncthreads();
  if (tid < 64) {
    const int cv = tid / 8, e = tid % 8;
    float s = 0.f;
    for (int t = cv; t < NT; t += 8)
      s += redf[t * 8 + e];
    P.sb2[(long)blockIdx.x * 64 + tid] = s;
  }
}

} // namespace aleppo
