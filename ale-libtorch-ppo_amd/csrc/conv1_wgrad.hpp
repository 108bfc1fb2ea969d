// conv1 weight gradient (bf16), tap-shift formulation.   dW1[oc][kh][kw][c] = sum_{n,oy,ox} dz1[n][oy][ox][oc] * obs[n][4oy+kh][4ox+kw][c]
//
// The implicit-GEMM form (rounds 1-2, an instantiation of conv_wgrad_patch_kernel: M = 32 channels, N = 256 taps, K = pixels; deleted in round 3) read one
// im2col fragment of the staged frame stack from LDS per TWO matrix instructions (there are only two 16-channel atoms
// to use it for) and was bound by exactly that: its consumer waves moved 205 KB of fragments per half sample through an LDS
// port that delivers 128 B per clock to `ds_read_b64_tr_b16` - 1,600 of the ~1,650 clocks the multiply phase takes alone
// on the CU (in-kernel `s_memtime` stamps; the staging waves never wait for HBM).  This kernel reads less:
//
//   kh = 4a + r, kw = 4b + s  (a, b in {0,1}; r, s in {0..3})   =>   obs[4oy+kh][4ox+kw][c] = X[oy+a][ox+b][(r,s,c)]
//
// where X is the 21 x 21 grid of 4x4-pixel cells of the frame (64 values per cell: exactly the bytes of a cell's four
// rows, no data movement).  With the output pixels indexed on the same 21-wide grid (column 20 = zeros),
//
//   dW[oc][(a,b)][(r,s,c)] = sum_k' DY[k' - 21a - b][oc] * X[k'][(r,s,c)]
//
// i.e. the four (a,b) blocks of the filter are the SAME X fragments multiplied by row-shifted views of the dY tile.  One
// wave now takes a whole k-step: 4 X fragments + 8 shifted dY fragments feed 32 matrix instructions (12 KB per 32
// instead of 24 KB per 32), and the four consumer waves split the k-steps (k-split inside the workgroup, summed once
// through LDS at the end, fixed order).  The bias gradient is one more matrix instruction per dY fragment (x ones).
//
// Roles: waves 0-3 prefetch (four half samples in flight in registers: the multiplying waves need none for them), widen
// and stage; waves 4-7 multiply.  The roles run SEPARATE loops with equal barrier counts: as branches inside one loop
// hipcc merges the wait-count state of the branch that issues the prefetch with the one that skips it and then waits for
// the set it has just requested (`s_waitcnt vmcnt(7)` where 15 is right).
#pragma once
#include "conv_patch.hpp"

namespace aleppo {
namespace c1w {
constexpr int IW = 84, C = 4, ROWS = 44, XROW = IW * C; // staged half frame: 44 rows of 84 pixels x 4 frames (bf16)
constexpr int X_ELEMS = ROWS * XROW;                     // 14,784
constexpr int XV = ROWS * IW * C / 16, DV = 200 * 32 / 8; // 16-byte source vectors per half sample: frame bytes, dY
constexpr int GSTRIDE = 40 * IW * C;                     // second half starts 40 rows down (bytes of packed u8)
constexpr int GW = 21;                                   // grid pitch: 20 output columns + one zero column
constexpr int KS = 8;                                    // k-steps of 32 grid cells (11 rows x 21 = 231 <= 256)
constexpr int KMAX = 11 * GW - 1;                        // last grid cell inside the staged half frame
constexpr int QOFF = 32;                                 // zero rows in front of the dY tile (>= 22, the largest shift)
constexpr int DYROWS = QOFF + KS * 32, DYS = 48;         // 32 channels + 16: conflict-free transposed reads (KFrag<bf16>)
constexpr int DY_ELEMS = DYROWS * DYS;
constexpr int BUF_ELEMS = X_ELEMS + DY_ELEMS;            // 28,608 bf16 per buffer
constexpr int NPW = 4, NPROD = NPW * 64, NTHREADS = NPROD + 256; // staging waves; + four multiplying waves
constexpr int NXV = (XV + NPROD - 1) / NPROD, NDV = (DV + NPROD - 1) / NPROD;
constexpr int NT = 4 * 2 * 4 + 2;                        // accumulator tiles per consumer wave: (a,b) x channel atom x r, + bias
constexpr size_t SMEM_LOOP = (size_t)2 * BUF_ELEMS * 2, SMEM_RED = (size_t)4 * NT * 64 * 16;
constexpr size_t SMEM = SMEM_LOOP > SMEM_RED ? SMEM_LOOP : SMEM_RED;
static_assert(X_ELEMS % 8 == 0 && BUF_ELEMS % 8 == 0, "16-byte aligned tiles");
} // namespace c1w

__global__ __launch_bounds__(c1w::NTHREADS) void conv1_wgrad_shift_kernel(WgradParams P) {
  using namespace c1w;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16 *sbuf = reinterpret_cast<bf16 *>(smem);
  typedef __attribute__((address_space(3))) bf16x4 *lds4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cw = (wave - NPW) & 3;
  const int li = lane & 15, lg = lane >> 4;
  const long ngroups = P.ns * 2, gs = gridDim.x; // group = half sample: dY rows [200 g, 200 g + 200)
  long grp = blockIdx.x;

  // every dY tile row that never receives data (the rows in front, the zero column, everything behind row 209) stays zero
  auto zero_dy_tiles = [&]() {
    for (int b = 0; b < 2; ++b)
      for (int e = tid; e < DY_ELEMS / 8; e += NTHREADS)
        reinterpret_cast<u32x4 *>(sbuf + (size_t)b * BUF_ELEMS + X_ELEMS)[e] = zero16();
    __syncthreads();
  };

  f32x4 acc[4][2][4], accb[2];
  if (wave < NPW) {
    // ------------------------------------------------------------------------------------ staging waves
    struct Regs {
      u32x4 x[NXV], d[NDV];
    };
    Regs R0, R1, R2, R3;
    // unconditional, clamped loads: hipcc then counts what is outstanding and waits for ONE set (conv_patch.hpp).
    // Non-temporal: this kernel runs beside the slab reduce of the other stream, and with ordinary loads its 220 MB
    // stream pushes the slabs the reduce is about to read out of L2 / the memory-side cache (same box: conv1 wgrad 88 ->
    // 74 us, the reduce 58 -> 46 us, update 7.70 -> 7.50 ms).  The hint is not a general win: on the operands of the other
    // conv kernels - cold or warm - it measured 3-8 us SLOWER per kernel (update 7.57-7.70 ms) and stays off there.
    auto gload = [&](Regs &R, long g) {
      g = min(g, ngroups - 1);
      const uint32_t n = (uint32_t)(g >> 1) + (uint32_t)P.map.n0, tp = (uint32_t)P.map.TP;
      const uint32_t q = n / tp, r = n - q * tp;
      const long off = (long)q * P.map.s1 + (long)r * P.map.s0 + P.map.base;
      const u32x4 *px = reinterpret_cast<const u32x4 *>(static_cast<const uint8_t *>(P.x) + off * 4 + (g & 1) * (long)GSTRIDE);
      const u32x4 *pd = reinterpret_cast<const u32x4 *>(P.dy + g * (long)(200 * 32));
#pragma unroll
      for (int i = 0; i < NXV; ++i)
        R.x[i] = __builtin_nontemporal_load(px + min(tid + NPROD * i, XV - 1));
#pragma unroll
      for (int i = 0; i < NDV; ++i)
        R.d[i] = __builtin_nontemporal_load(pd + min(tid + NPROD * i, DV - 1));
    };
    auto swrite = [&](Regs &R, int buf) {
      bf16 *dx = sbuf + (size_t)buf * BUF_ELEMS, *dd = dx + X_ELEMS;
#pragma unroll
      for (int i = 0; i < NXV; ++i) {
        const int v = tid + NPROD * i;
        if (v < XV) {
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const uint32_t w0 = R.x[i][2 * d], w1 = R.x[i][2 * d + 1];
            reinterpret_cast<u32x4 *>(dx)[2 * v + d] =
                u32x4{pack_u8_pair_bf16(w0 & 255u, (w0 >> 8) & 255u), pack_u8_pair_bf16((w0 >> 16) & 255u, w0 >> 24),
                      pack_u8_pair_bf16(w1 & 255u, (w1 >> 8) & 255u), pack_u8_pair_bf16((w1 >> 16) & 255u, w1 >> 24)};
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NDV; ++i) {
        const int v = tid + NPROD * i;
        if (v < DV) {
          const int p = v >> 2, cv = v & 3; // output pixel oy * 20 + ox -> grid row oy * 21 + ox
          *reinterpret_cast<u32x4 *>(dd + (QOFF + p + p / 20) * DYS + cv * 8) = R.d[i];
        }
      }
    };
    gload(R0, grp); // (requested before anything else: the first round trip to HBM is the kernel's start-up time)
    gload(R1, grp + gs);
    gload(R2, grp + 2 * gs);
    gload(R3, grp + 3 * gs);
    zero_dy_tiles();
    swrite(R0, 0);
    gload(R0, grp + 4 * gs);
    __syncthreads();
    for (; grp < ngroups; grp += 4 * gs) { // (past the end the clamped re-read of the last group is staged: not multiplied)
      swrite(R1, 1);
      gload(R1, grp + 5 * gs);
      __syncthreads();
      swrite(R2, 0);
      gload(R2, grp + 6 * gs);
      __syncthreads();
      swrite(R3, 1);
      gload(R3, grp + 7 * gs);
      __syncthreads();
      swrite(R0, 0);
      gload(R0, grp + 8 * gs);
      __syncthreads();
    }
  } else {
    // ------------------------------------------------------------------------------------ multiplying waves
#pragma unroll
    for (int sh = 0; sh < 4; ++sh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[sh][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // wave cw owns k-steps cw and cw + 4.  X operand: transposed read h of a k-step supplies grid cell
    // 32 t + 16 h + 4 lg + (li >> 2), columns 4 (li & 3) .. + 3 of the 16 (s, c) values of cell row r.
    int xoff[2][2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = min(32 * (cw + 4 * tl) + 16 * h + 4 * lg + (li >> 2), KMAX); // (cells past the frame meet zero dY rows)
        const int gy = k / GW, gx = k - gy * GW;
        xoff[tl][h] = (4 * gy * IW + 4 * gx) * C + 4 * (li & 3);
      }
    const int aoff = (32 * cw + 4 * lg + (li >> 2)) * DYS + 4 * (li & 3); // dY operand: row of this lane in k-step cw
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    auto multiply = [&](int buf) {
      const bf16 *px = sbuf + (size_t)buf * BUF_ELEMS, *pd = px + X_ELEMS + aoff;
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        u32x4 fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(px + xoff[tl][0] + j * XROW)));
          const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(px + xoff[tl][1] + j * XROW)));
          fb[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int sh = 0; sh < 4; ++sh) {
          const int row = QOFF + 128 * tl - (sh >> 1) * GW - (sh & 1); // view of the dY tile shifted by (a, b) = (sh >> 1, sh & 1)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(pd + row * DYS + 16 * i)));
            const u32x2 hi =
                __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(pd + (row + 16) * DYS + 16 * i)));
            const u32x4 fa = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int j = 0; j < 4; ++j)
              Atom<bf16>::mma(fa, fb[j], acc[sh][i][j]);
            if (sh == 0)
              Atom<bf16>::mma(fa, ones, accb[i]); // bias gradient: column sums of the unshifted tile
          }
        }
      }
    };
    zero_dy_tiles();
    __syncthreads();
    for (; grp < ngroups; grp += 4 * gs) {
      multiply(0);
      __syncthreads();
      if (grp + gs < ngroups)
        multiply(1);
      __syncthreads();
      if (grp + 2 * gs < ngroups)
        multiply(0);
      __syncthreads();
      if (grp + 3 * gs < ngroups)
        multiply(1);
      __syncthreads();
    }
  }
  // ---- the four k-split parts are added in wave order; one slab per workgroup: [32][256] weights + [32] bias
  f32x4 *red = reinterpret_cast<f32x4 *>(smem);
  if (wave >= NPW) {
#pragma unroll
    for (int sh = 0; sh < 4; ++sh)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          red[(cw * NT + (sh * 2 + i) * 4 + j) * 64 + lane] = acc[sh][i][j];
    red[(cw * NT + 32) * 64 + lane] = accb[0];
    red[(cw * NT + 33) * 64 + lane] = accb[1];
  }
  __syncthreads();
  float *ow = P.slab_w + (long)blockIdx.x * 32 * 256, *ob = P.slab_b + (long)blockIdx.x * 32;
  for (int t = wave; t < NT; t += NTHREADS / 64) {
    f32x4 v = red[t * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 u = red[(w * NT + t) * 64 + lane];
      v = f32x4{v[0] + u[0], v[1] + u[1], v[2] + u[2], v[3] + u[3]};
    }
    if (t < 32) { // tile ((a,b), channel atom i, cell row r): columns (kh = 4a + r, kw = 4b + s, c), li = 4 s + c
      const int sh = t >> 3, i = (t >> 2) & 1, r = t & 3;
      const int col = ((4 * (sh >> 1) + r) * 8 + 4 * (sh & 1)) * 4 + li;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ow[(long)(i * 16 + lg * 4 + e) * 256 + col] = v[e] * P.scale;
    } else if (li == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ob[(t - 32) * 16 + lg * 4 + e] = v[e];
    }
  }
}

} // namespace aleppo
