// conv3_bwd_fused.hpp - conv3's data gradient AND weight gradient in one launch (bf16).
//
//   dz2[y][x][c]  = (a2[y][x][c] > 0) * sum_{kh,kw,oc} W3[oc][kh][kw][c] * dz3[y-kh][x-kw][oc]          (9 x 9 x 64 per sample)
//   dW3[oc][kh][kw][c] += sum_{oy,ox} dz3[oy][ox][oc] * a2[oy+kh][ox+kw][c],   db3[oc] += sum dz3[.][.][oc]
//
// Both read the same two tensors of a sample (dz3 6 KB, a2 10 KB); as two launches (conv3_dgrad_tile_kernel +
// conv_wgrad_patch_kernel<LConv3Wgrad>, 37 + 42 us alone per 4096-sample minibatch) they fetch them twice and keep one
// LDS read in flight per MFMA pair.  Here a workgroup (4 waves, one per SIMD, the whole register file each; persistent over
// samples, conv_fwd_fused.hpp's scheme) stages a sample ONCE into LDS and runs
//   phase 1 (dgrad): 6 pixel atoms x 2 channel groups, W3d (this wave's 32 channels x 576) resident in 144 AGPRs, fragments
//            from a ZERO-BORDERED 11 x 11 image of dz3 (no tap masking), ReLU gate + bf16 + one 16-byte global store per
//            pixel atom in the MFMA gaps of the next atom;
//   phase 2 (wgrad): this wave's quarter of dW3 (4 channel atoms x 9 taps x 16 input channels = 36 accumulator tiles, 144
//            VGPRs, summed over ALL samples of the workgroup: one fp32 slab per workgroup at the end, as before), both
//            operands by `ds_read_b64_tr_b16` from pixel-major tiles;
// with the LDS reads and MFMAs of both phases spelled out as software pipelines (inline assembly, hand-counted waits).
// LDS is double buffered (2 x 43 KB): sample n + 1 is staged from registers requested a whole iteration earlier while sample
// n is being multiplied - ONE barrier per sample.
#pragma once
#include "conv_fwd_fused.hpp"

namespace aleppo {

struct Conv3BwdParams {
  const bf16 *dz3, *a2, *w3d; // [ns][49][64], [ns][81][64], [64 c][9 taps][64 oc]
  bf16 *dz2;                  // [ns][81][64]
  float *slab_w, *slab_b;     // [grid][64][576], [grid][64]
  long ns;
};

namespace c3b {
constexpr int NT = 256;
// dz3, zero-bordered (dgrad B operand, ds_read_b128): 11 x 11 pixels, pixel pitch 72 elements (144 B: the 16 lanes of a read
// group - consecutive output pixels - fall on 16 different 16-byte bank slots), row pitch 904 (= 9 pixels further modulo
// 128 elements: the pattern continues across the end of a 9-pixel output row)
constexpr int CP = 72, PR = 904, IMG_ELEMS = 11 * PR;
// dz3, pixel-major (wgrad A operand, transposed reads): 64 rows (49 pixels + zero rows) x (64 channels + 16)
constexpr int DYS = 80, TILE_ELEMS = 64 * DYS;
// a2 (wgrad B operand, transposed reads; ReLU gates): 81 pixels x (64 channels + 16)
constexpr int XP = 80, X_ELEMS = 81 * XP;
constexpr int BUF_ELEMS = IMG_ELEMS + TILE_ELEMS + X_ELEMS; // 21,528 bf16 = 43,056 B
constexpr size_t SMEM = (size_t)2 * BUF_ELEMS * 2;
constexpr int DV = 49 * 8, XV = 81 * 8; // 16-byte source vectors per sample
constexpr int KS = 18;                  // dgrad k-steps: 9 taps x 64 channels
static_assert(BUF_ELEMS % 8 == 0 && IMG_ELEMS % 8 == 0 && TILE_ELEMS % 8 == 0, "16-byte aligned tiles");
constexpr int dgrad_koff(int ks) { // byte offset of k-step ks from output pixel (y, x)'s own cell of the bordered image
  const int tap = ks / 2, dy = tap / 3, dx = tap % 3;
  return 2 * ((2 - dy) * PR + (2 - dx) * CP + (ks % 2) * 32);
}
} // namespace c3b

template <int OFF> __device__ __forceinline__ void ff_read_tr(u32x2 &dst, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void ff_mfma_vv(f32x4 &acc, const u32x2 &alo, const u32x2 &ahi, const u32x2 &blo, const u32x2 &bhi) {
  const u32x4 a = {alo[0], alo[1], ahi[0], ahi[1]}, b = {blo[0], blo[1], bhi[0], bhi[1]};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// vmcnt(NVM): the vector-memory operations the thread issued behind its prefetch loads (which wrote these AGPRs)
template <int NVM> __device__ __forceinline__ void ff_wait_prefetch5(u32x4 &a, u32x4 &b, u32x4 &c, u32x4 &d, u32x4 &e) {
  asm volatile("s_waitcnt vmcnt(%5)" : "+a"(a), "+a"(b), "+a"(c), "+a"(d), "+a"(e) : "n"(NVM) : "memory");
}

// ABL: timing-only ablations (wrong results): 1 no dgrad phase, 2 no wgrad phase, 4 no slab write
template <int ABL>
__global__ __launch_bounds__(c3b::NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3_bwd_fused_kernel(Conv3BwdParams P) {
  using namespace c3b;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  typedef __attribute__((address_space(3))) uint8_t *lds_ptr;
  bf16 *sbuf = reinterpret_cast<bf16 *>(smem);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4, li = fr, lg = fg;
  const int og = wave & 1, pl = wave >> 1; // dgrad: channel group (32 of the 64 input channels), pixel lane
  const long gs = gridDim.x;
  long n = blockIdx.x;
  if (n >= P.ns)
    return;

  // ---- dgrad weights: W3d rows 32 og + (fr >> 2) * 8 + a * 4 + (fr & 3) (FusedW's row mapping: 8 consecutive channels per lane)
  u32x4 W[2][KS];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      W[a][ks] = *reinterpret_cast<const u32x4 *>(P.w3d + (long)(og * 32 + (fr >> 2) * 8 + a * 4 + (fr & 3)) * 576 + ks * 32 +
                                                  fg * 8);
  // ---- wgrad accumulators: tile (i, j) = channel atom i x (tap j, input channels 16 wave .. + 15)
  f32x4 acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j)
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // this thread's 8 channels (tid & 7) of db3

  // ---- zero both buffers once: image border, tile rows 49..63 (and everything else) are never written afterwards
  for (int e = tid; e < 2 * BUF_ELEMS / 8; e += NT)
    reinterpret_cast<u32x4 *>(sbuf)[e] = zero16();

  // ---- prefetch registers (AGPRs, inline-assembly loads): 2 dz3 vectors + 3 a2 vectors per thread and sample
  u32x4 RD[2], RX[3];
  auto prefetch = [&](long m) { // unconditional, clamped
    const long mm = min(m, P.ns - 1);
    const u32x4 *pd = reinterpret_cast<const u32x4 *>(P.dz3 + mm * (49 * 64));
    const u32x4 *px = reinterpret_cast<const u32x4 *>(P.a2 + mm * (81 * 64));
#pragma unroll
    for (int i = 0; i < 2; ++i)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(RD[i]) : "v"(pd + min(tid + NT * i, DV - 1)) : "memory");
#pragma unroll
    for (int i = 0; i < 3; ++i)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(RX[i]) : "v"(px + min(tid + NT * i, XV - 1)) : "memory");
  };
  auto stage = [&](int b, bool counts) { // counts: a real sample (not the clamped re-read past the end): part of db3
    bf16 *img = sbuf + (size_t)b * BUF_ELEMS, *tile = img + IMG_ELEMS, *xs = tile + TILE_ELEMS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + NT * i;
      if (v < DV) {
        const int p = v >> 3, part = v & 7, py = p / 7, px = p - py * 7;
        *reinterpret_cast<u32x4 *>(img + (py + 2) * PR + (px + 2) * CP + part * 8) = RD[i];
        *reinterpret_cast<u32x4 *>(tile + p * DYS + part * 8) = RD[i];
        if (counts)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bsum[2 * e] += bf16_bits_to_f32(RD[i][e] & 0xFFFFu);
            bsum[2 * e + 1] += bf16_bits_to_f32(RD[i][e] >> 16);
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int v = tid + NT * i;
      if (v < XV)
        *reinterpret_cast<u32x4 *>(xs + (v >> 3) * XP + (v & 7) * 8) = RX[i];
    }
  };

  // ---- lane constants of the two phases (buffer 0; the other buffer is + BUF_ELEMS * 2 bytes)
  uint32_t dbase[3], gaddr[3];
  int qv[3];
#pragma unroll
  for (int ak = 0; ak < 3; ++ak) {
    const int q = min((pl + ak * 2) * 16 + fr, 80), y = q / 9, x = q - y * 9;
    qv[ak] = q;
    dbase[ak] = lds0 + 2u * (uint32_t)(y * PR + x * CP + fg * 8);
    gaddr[ak] = lds0 + 2u * (uint32_t)(IMG_ELEMS + TILE_ELEMS + q * XP + og * 32 + fg * 8);
  }
  // wgrad: transposed read h of k-step ks supplies pixel row 32 ks + 16 h + 4 lg + (li >> 2) (rows past 48 are zero in the dz3
  // tile: their a2 address is clamped), columns 4 (li & 3) .. + 3 of the atom
  uint32_t abase, xbase[2][2];
  abase = lds0 + 2u * (uint32_t)(IMG_ELEMS + (4 * lg + (li >> 2)) * DYS + 4 * (li & 3));
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = min(ks * 32 + h * 16 + 4 * lg + (li >> 2), 48), oy = q / 7, ox = q - oy * 7;
      xbase[ks][h] = lds0 + 2u * (uint32_t)(IMG_ELEMS + TILE_ELEMS + (oy * 9 + ox) * XP + wave * 16 + 4 * (li & 3));
    }

  __syncthreads(); // zeroed
  prefetch(n);
  ff_wait_prefetch5<0>(RD[0], RD[1], RX[0], RX[1], RX[2]);
  stage(0, true);
  prefetch(n + gs);
  __syncthreads();

  for (int b = 0; n < P.ns; n += gs, b ^= 1) {
    const uint32_t boff = b ? (uint32_t)(BUF_ELEMS * 2) : 0u;
    // =========================================================== phase 1: dz2 of sample n
    if constexpr (!(ABL & 1)) {
      constexpr int APW = 3, N = APW * KS, D = 8;
      u32x4 frag[D], gate[APW];
      f32x4 dacc[2][2];
      uint32_t packed[4];
      uint32_t base[APW];
#pragma unroll
      for (int ak = 0; ak < APW; ++ak) {
        base[ak] = dbase[ak] + boff;
        asm volatile("ds_read_b128 %0, %1" : "=v"(gate[ak]) : "v"(gaddr[ak] + boff)); // (older than every fragment read)
      }
      bf16 *gout = P.dz2 + n * (long)(81 * 64) + og * 32 + fg * 8;
      auto issue = [&](auto I) {
        constexpr int i = decltype(I)::value, ak = i / KS, ks = i % KS;
        ff_read<dgrad_koff(ks)>(frag[i % D], base[ak]);
      };
      auto epilogue = [&](auto AK, auto PIECE) {
        constexpr int ak = decltype(AK)::value, piece = decltype(PIECE)::value, set = ak & 1;
        if constexpr (piece < 4) { // channels 2 piece, 2 piece + 1: gated by the stored bf16 activation
          const f32x4 &a = dacc[set][piece >> 1];
          constexpr int r = (piece & 1) * 2;
          const uint32_t g = gate[ak][piece];
          const float v0 = bf16_bits_to_f32(g & 0xFFFFu) > 0.f ? a[r] : 0.f, v1 = bf16_bits_to_f32(g >> 16) > 0.f ? a[r + 1] : 0.f;
          packed[piece] = pack2_bf16(v0, v1);
        } else {
          *reinterpret_cast<u32x4 *>(gout + qv[ak] * 64) = u32x4{packed[0], packed[1], packed[2], packed[3]};
        }
      };
      static_for<D - 1>(issue);
      static_for<N>([&](auto I) {
        constexpr int i = decltype(I)::value, ak = i / KS, ks = i % KS, set = ak & 1;
        if constexpr (i + D - 1 < N)
          issue(IC<i + D - 1>{});
        ff_wait_lgkm<(N - 1 - i < D - 1 ? N - 1 - i : D - 1)>();
        if constexpr (i == 0) // the gates are older than every fragment read: landed; ordinary code may name them from here on
          asm volatile("" : "+v"(gate[0]), "+v"(gate[1]), "+v"(gate[2]));
        if constexpr (ks == 0) {
          ff_mfma0<true>(dacc[set][0], W[0][0], frag[i % D]);
          ff_mfma0<true>(dacc[set][1], W[1][0], frag[i % D]);
        } else {
          ff_mfma<true>(dacc[set][0], W[0][ks], frag[i % D]);
          ff_mfma<true>(dacc[set][1], W[1][ks], frag[i % D]);
        }
        if constexpr (ak > 0 && ks >= 1 && ks <= 6) {
          if constexpr (ks == 1)
            ff_release(dacc[set ^ 1][0], dacc[set ^ 1][1]);
          else
            epilogue(IC<ak - 1>{}, IC<ks - 2>{});
        }
      });
      ff_release_nops(dacc[(APW - 1) & 1][0], dacc[(APW - 1) & 1][1]);
      static_for<5>([&](auto PIECE) { epilogue(IC<APW - 1>{}, PIECE); });
    }
    // =========================================================== phase 2: dW3 += sample n
    if constexpr (!(ABL & 2)) {
      constexpr int NI = 18, DB = 4; // items (k-step, tap); ring of DB B fragments: DB - 1 items (2 reads each) in flight
      u32x2 fa[2][4][2], fb[DB][2];
      const uint32_t ab = abase + boff;
      uint32_t xb[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          xb[ks][h] = xbase[ks][h] + boff;
      // A operand (dz3 tile) of both k-steps: 16 reads, older than every B read
      static_for<2>([&](auto KSI) {
        static_for<4>([&](auto II) {
          constexpr int ks = decltype(KSI)::value, i = decltype(II)::value;
          ff_read_tr<2 * (ks * 32 * DYS + i * 16)>(fa[ks][i][0], ab);
          ff_read_tr<2 * ((ks * 32 + 16) * DYS + i * 16)>(fa[ks][i][1], ab);
        });
      });
      auto issue = [&](auto T) {
        constexpr int t = decltype(T)::value, ks = t / 9, j = t % 9;
        constexpr int joff = 2 * (((j / 3) * 9 + (j % 3)) * XP);
        ff_read_tr<joff>(fb[t % DB][0], xb[ks][0]);
        ff_read_tr<joff>(fb[t % DB][1], xb[ks][1]);
      };
      static_for<DB - 1>(issue);
      static_for<NI>([&](auto T) {
        constexpr int t = decltype(T)::value, ks = t / 9, j = t % 9;
        if constexpr (t + DB - 1 < NI)
          issue(IC<t + DB - 1>{});
        // (the wait names the registers it covers: the 64-bit halves are merged into 128-bit operands by ordinary code,
        // which must not move in front of it)
        if constexpr (j == 0)
          asm volatile("s_waitcnt lgkmcnt(%10)"
                       : "+v"(fb[t % DB][0]), "+v"(fb[t % DB][1]), "+v"(fa[ks][0][0]), "+v"(fa[ks][0][1]), "+v"(fa[ks][1][0]),
                         "+v"(fa[ks][1][1]), "+v"(fa[ks][2][0]), "+v"(fa[ks][2][1]), "+v"(fa[ks][3][0]), "+v"(fa[ks][3][1])
                       : "n"(2 * (NI - 1 - t < DB - 1 ? NI - 1 - t : DB - 1)));
        else
          asm volatile("s_waitcnt lgkmcnt(%2)"
                       : "+v"(fb[t % DB][0]), "+v"(fb[t % DB][1])
                       : "n"(2 * (NI - 1 - t < DB - 1 ? NI - 1 - t : DB - 1)));
        static_for<4>([&](auto II) {
          constexpr int i = decltype(II)::value;
          ff_mfma_vv(acc[i][j], fa[ks][i][0], fa[ks][i][1], fb[t % DB][0], fb[t % DB][1]);
        });
      });
    }
    // =========================================================== stage sample n + gs into the other buffer
    ff_wait_prefetch5<(ABL & 1) ? 0 : 3>(RD[0], RD[1], RX[0], RX[1], RX[2]); // behind the prefetch: the 3 dz2 stores of phase 1
    stage(b ^ 1, n + gs < P.ns);
    prefetch(n + 2 * gs);
    __syncthreads();
  }

  // ---- one slab per workgroup
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); // (the last MFMAs' results are read by ordinary code below)
  float *ow = P.slab_w + (long)blockIdx.x * 64 * 576;
  if constexpr (!(ABL & 4))
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      f32x4 v = acc[i][j];
      asm volatile("" : "+v"(v));
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ow[(long)(i * 16 + lg * 4 + r) * 576 + j * 64 + wave * 16 + li] = v[r];
    }
  // bias: threads with equal tid & 7 hold the same 8 channels; ordered LDS reduction (deterministic)
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    red[tid * 8 + e] = bsum[e];
  __syncthreads();
  if (tid < 64) {
    const int cv = tid / 8, e = tid % 8;
    float s = 0.f;
    for (int t = cv; t < NT; t += 8)
      s += red[t * 8 + e];
    P.slab_b[(long)blockIdx.x * 64 + tid] = s;
  }
}

} // namespace aleppo
