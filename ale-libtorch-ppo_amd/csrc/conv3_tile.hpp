// conv3_tile.hpp - conv3 dgrad as a register-tiled implicit GEMM out of LDS (bf16).
//
//   dz2[n][y][x][c] = relu'(a2) * sum_{dy,dx,oc} dz3[n][y-dy][x-dx][oc] * W3[oc][c][dy][dx]      (3x3, stride 1)
//
// The sample-stationary kernel (conv_patch.hpp) keeps the 18 k-steps of weights of two channel atoms in registers
// (144 VGPRs): nothing is left for fragments in flight, so every atom is a serialized LDS-read -> 2 MFMA chain (PMC:
// 22 % MFMA busy, 45 us).  Here the WEIGHTS live in LDS (74 KB, staged once per persistent workgroup) and a wave
// owns a 3-pixel-atom x 4-channel-atom tile (48 pixels x 64 channels, 48 accumulator VGPRs): per k-step 3 pixel
// fragments + 4 weight fragments feed 12 MFMAs, and with ~130 VGPRs free the fully unrolled k-loop keeps the next
// fragments in flight.  A group is 4 samples = 324 output pixels = 21 atoms = 7 waves x 3 atoms (the 8th wave runs
// the same instruction stream on clamped data so that every wave issues the same number of vector-memory
// operations: counted waits, see conv_patch.hpp).  Weight rows are stored atom-major (row a*16 + i holds channel
// (i >> 2)*16 + a*4 + (i & 3)) with a 1184-byte pitch (conflict-free ds_read_b128, tests/tools/lds_conflicts.py), so a
// lane ends up with 16 CONSECUTIVE channels: the epilogue is two 16-byte pieces per pixel and the 4 lanes of a
// pixel write its whole 128-byte line.
#pragma once
#include "conv_patch.hpp"

namespace aleppo {

struct C3TileParams {
  const bf16 *dy;   // dz3 [ns][7][7][64]
  const bf16 *w;    // W3d [64 c][576 = (dy, dx, oc)]
  const bf16 *gate; // a2 [ns][9][9][64]
  bf16 *out;        // dz2 [ns][9][9][64]
  bf16 *dummy;      // >= 16 KB scratch for lanes without a pixel
  long ns;
};

constexpr int C3T_SB = 4;                       // samples per group
constexpr int C3T_CP = 72, C3T_RP = 7 * 72 + 80; // patch pixel / row pitch (elements), as LConv3Dgrad
constexpr int C3T_SP = 7 * C3T_RP;              // sample pitch
constexpr int C3T_WP = 1184 / 2;                // weight row pitch (elements)
constexpr size_t C3T_SMEM = (size_t)64 * C3T_WP * 2 + (size_t)2 * C3T_SB * C3T_SP * 2;

__global__ __launch_bounds__(512) void conv3_dgrad_tile_kernel(C3TileParams P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16 *sw = reinterpret_cast<bf16 *>(smem);
  bf16 *sp = sw + 64 * C3T_WP;
  constexpr int BUF = C3T_SB * C3T_SP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const long ngroups = (P.ns + C3T_SB - 1) / C3T_SB;

  // ---- weights -> LDS once (4608 chunks, 9 per thread), rows permuted atom-major
  for (int v = tid; v < 64 * 72; v += 512) {
    const int c = v / 72, ch = v - c * 72;
    const int row = ((c >> 2) & 3) * 16 + (c >> 4) * 4 + (c & 3); // channel c = (i>>2)*16 + a*4 + (i&3) -> row a*16 + i
    *reinterpret_cast<u32x4 *>(sw + row * C3T_WP + ch * 8) = reinterpret_cast<const u32x4 *>(P.w)[v];
  }

  // ---- patch prefetch: 4 samples x 392 chunks = 1568 chunks, 4 (clamped, unconditional) loads per thread
  u32x4 R[4];
  auto gload = [&](long grp) {
    const long n0 = min(grp, ngroups - 1) * C3T_SB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = min(tid + 512 * i, C3T_SB * 392 - 1);
      const long n = min(n0 + v / 392, P.ns - 1);
      R[i] = reinterpret_cast<const u32x4 *>(P.dy + n * 3136)[v % 392];
    }
  };
  auto swrite = [&](int buf, long grp) {
    bf16 *dst = sp + (size_t)buf * BUF;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 512 * i;
      if (v < C3T_SB * 392) {
        const int u = v / 392, w = v - u * 392, px = w >> 3, c = w & 7, row = px / 7, col = px - row * 7;
        if (grp * C3T_SB + u >= P.ns)
          R[i] = zero16();
        *reinterpret_cast<u32x4 *>(dst + u * C3T_SP + row * C3T_RP + col * C3T_CP + c * 8) = R[i];
      }
    }
  };

  // this wave's 3 pixel atoms: pixel q = (3 wave + t) * 16 + fr of the group's 324 (wave 7: none)
  auto group = [&](int buf, long grp) {
    const bf16 *pb = sp + (size_t)buf * BUF;
    const long n0 = grp * C3T_SB;
    const int count = (int)max(0L, min((long)C3T_SB, P.ns - n0)); // samples of this group
    int py[3], px[3], pbase[3];
    bool ok[3];
    long off[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int q = (3 * wave + t) * 16 + fr;
      ok[t] = wave < 7 && q < count * 81;
      const int s = min(q / 81, C3T_SB - 1), p = q - (q / 81) * 81;
      py[t] = p / 9;
      px[t] = p - py[t] * 9;
      pbase[t] = s * C3T_SP + fg * 8;
      off[t] = ((n0 + s) * 81 + p) * 64L + fg * 16;
    }
    // ReLU gates of the 3 pixels (16 channels = 32 B per lane), requested before the MFMA chain
    u32x4 g0[3], g1[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bf16 *gp = P.gate + (ok[t] ? off[t] : 0);
      g0[t] = *reinterpret_cast<const u32x4 *>(gp);
      g1[t] = *reinterpret_cast<const u32x4 *>(gp + 8);
    }
    f32x4 acc[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int a = 0; a < 4; ++a)
        acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 18; ++ks) {
      const int tap = ks >> 1, dy = tap / 3, dx = tap - dy * 3;
      u32x4 fw[4], fx[3];
#pragma unroll
      for (int a = 0; a < 4; ++a)
        fw[a] = *reinterpret_cast<const u32x4 *>(sw + (a * 16 + fr) * C3T_WP + ks * 32 + fg * 8);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int sy = py[t] - dy, sx = px[t] - dx;
        const bool in = ok[t] && sy >= 0 && sy < 7 && sx >= 0 && sx < 7;
        const int cy = min(max(sy, 0), 6), cx = min(max(sx, 0), 6); // clamped: a neighbour's address (broadcast)
        u32x4 b = *reinterpret_cast<const u32x4 *>(pb + pbase[t] + cy * C3T_RP + cx * C3T_CP + (ks & 1) * 32);
        fx[t] = in ? b : zero16(); // (reading a zero block instead of selecting the data measured no faster)
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          Atom<bf16>::mma(fw[a], fx[t], acc[t][a]);
    }
    // epilogue: lane (fr = pixel, fg) holds channels fg*16 + a*4 + r
    auto gated = [](uint32_t w, int h, float v) {
      const float g = bf16_bits_to_f32(h ? (w >> 16) : (w & 0xFFFFu));
      return g > 0.f ? v : 0.f;
    };
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      u32x2 o[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const u32x4 &g = a < 2 ? g0[t] : g1[t];
        const int b = (a & 1) * 2;
        o[a] = pack4_bf16(gated(g[b], 0, acc[t][a][0]), gated(g[b], 1, acc[t][a][1]), gated(g[b + 1], 0, acc[t][a][2]),
                          gated(g[b + 1], 1, acc[t][a][3]));
      }
      bf16 *dst = ok[t] ? P.out + off[t] : P.dummy + tid * 16;
      *reinterpret_cast<u32x4 *>(dst) = u32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
      *reinterpret_cast<u32x4 *>(dst + 8) = u32x4{o[2][0], o[2][1], o[3][0], o[3][1]};
    }
  };

  const long gs = gridDim.x;
  long grp = blockIdx.x;
  gload(grp);
  swrite(0, grp);
  gload(grp + gs);
  __syncthreads();
  for (int it = 0; grp < ngroups; grp += gs, ++it) {
    group(it & 1, grp);
    swrite((it + 1) & 1, grp + gs);
    gload(grp + 2 * gs);
    __syncthreads();
  }
}

} // namespace aleppo
