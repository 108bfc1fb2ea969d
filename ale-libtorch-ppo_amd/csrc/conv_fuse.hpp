// conv_fuse.hpp - conv2 data gradient FUSED with conv1's weight gradient (bf16 training path, gfx950).
//
// autograd's backward of NetworkImpl::forward (reference src/bin/train.cc:232-236, 255-261; ai::ppo::train::
// mini_batch_update, src/ai/ppo/train.h:126) computes, for the first two convolutions,
//     dz1 = relu'(a1) * conv_transpose(dz2, W2)                (conv2 dgrad, 21.7 GFLOP per 4096-sample minibatch)
//     dW1 = dz1^T * im2col(x / 255),  db1 = sum dz1            (conv1 wgrad, 26.8 GFLOP)
// As separate kernels dz1 (25.6 KB per sample, 105 MB per minibatch) is written to HBM by the first and read back by
// the second, and conv1's weight gradient is the last link of the update's dependency chain.  Here ONE persistent
// workgroup per CU streams samples: dz2 (10 KB), the packed uint8 observation (28 KB) and the ReLU gates a1 (25.6 KB)
// are the only HBM reads, dz1 goes from the MFMA accumulators through LDS straight into the weight-gradient MFMAs, and
// dW1 / db1 stay in registers across all samples of the workgroup (one fp32 slab per workgroup, as before).
// Per minibatch: 210 MB less HBM traffic, one launch less, and the two MFMA phases of a sample share its staging.
//
// Phase A (= conv_patch_kernel<LConv2Dgrad>): the 20x20 input pixels split into the 4 parity classes (py, px); a class
//   only sees the 2x2 taps kh = py (mod 2), kw = px (mod 2) of the 4x4 stride-2 kernel.  Wave w owns class w & 3 and
//   every second 16-pixel atom of its 10x10 pixel grid; the class's weights (32 channels x 256 k) live in registers.
//   The gated bf16 result lands in LDS as the [pixel][channel] tile phase B consumes.
// Phase B (= conv_wgrad_patch_kernel<LConv1Wgrad>, whole samples): dW1[oc][j] += sum_pixel dz1[pixel][oc] *
//   im2col(x)[pixel][j]; both operands k(=pixel)-outer in LDS, read with ds_read_b64_tr_b16.
#pragma once
#include "conv_patch.hpp"

namespace aleppo {

struct FuseC2dC1wParams {
  const bf16 *dz2;     // [ns][81][64]
  const bf16 *w2d;     // [4 classes][32][256]  (W2d: dgrad layout of conv2's weights)
  const bf16 *a1;      // [ns][400][32] forward activation of conv1 (ReLU gate)
  const uint8_t *obs;  // packed uint8 stacks, addressed through map
  SampleMap map;
  float *slab_w;       // [gridDim.x][32][256]
  float *slab_b;       // [gridDim.x][32]
  long ns;
  float scale;         // 1/255: the x/255 of train.cc:258-259 folded into the weight gradient
};

constexpr int F21_X_ELEMS = 84 * 84 * 4;                       // widened observation, [pixel][4] bf16
constexpr int F21_D2_ELEMS = PatchGeom<LConv2Dgrad>::SP;       // padded dz2 image (one sample)
constexpr int F21_DYS = 32 + 16;                               // dz1 tile row stride (elements): conflict-free tr reads
constexpr int F21_KS = 13;                                     // 400 pixels = 12.5 k-steps of 32: rows 400..415 stay zero
constexpr int F21_DY_ELEMS = F21_KS * 32 * F21_DYS;
constexpr size_t F21_SMEM = (size_t)(F21_X_ELEMS + F21_D2_ELEMS + F21_DY_ELEMS) * 2;

__global__ __launch_bounds__(512) void conv2_dgrad_conv1_wgrad_kernel(FuseC2dC1wParams P) {
  using LA = LConv2Dgrad;
  using GEO = PatchGeom<LA>;
  constexpr int RP = GEO::RP, CP = LA::CP;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  bf16 *sx = reinterpret_cast<bf16 *>(smem), *sd2 = sx + F21_X_ELEMS, *sdy = sd2 + F21_D2_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;

  // ---------------- phase A constants: class weights -> registers (A operand: row = channel, k = 32 ks + 8 fg)
  const int og = wave & 3, pl = wave >> 2; // parity class, pixel lane (atoms pl, pl + 2, ..)
  const int py = og >> 1, px = og & 1;
  constexpr int KA = 256, KSA = 8, NATOM = 7, APW = 4; // 100 pixels per class = 7 atoms, 4 / 3 per wave
  u32x4 W[2][KSA];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks) // row fr of atom a = channel (fr >> 2) * 8 + a * 4 + (fr & 3): a lane ends up with
                                     // 8 consecutive channels fg * 8 .. + 7 (one 16-byte piece)
      W[a][ks] = *reinterpret_cast<const u32x4 *>(P.w2d + (long)(og * 32 + (fr >> 2) * 8 + a * 4 + (fr & 3)) * KA +
                                                  ks * 32 + fg * 8);
  // this wave's atoms: pixel (y, x) of the class grid -> conv1-output pixel p1 = (2y + py) * 20 + 2x + px
  int p1[APW];
  bool qok[APW];
  int ya[APW], xa[APW];
#pragma unroll
  for (int ak = 0; ak < APW; ++ak) {
    const int q = (pl + ak * 2) * 16 + fr;
    qok[ak] = pl + ak * 2 < NATOM && q < 100;
    const int qq = min(q, 99);
    ya[ak] = qq / 10;
    xa[ak] = qq - ya[ak] * 10;
    p1[ak] = (2 * ya[ak] + py) * 20 + 2 * xa[ak] + px;
  }

  // ---------------- phase B constants (conv_wgrad_patch_kernel<LConv1Wgrad>, whole-sample units)
  const int wn = wave, li = lane & 15, lg = lane >> 4;
  int pixoff[F21_KS][2];
#pragma unroll
  for (int ks = 0; ks < F21_KS; ++ks)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int q = min(ks * 32 + r * 16 + 4 * lg + (li >> 2), 399); // (tail rows of the dz1 tile are zero)
      const int oy = q / 20, ox = q - oy * 20;
      pixoff[ks][r] = ((oy * 4) * 84 + ox * 4) * 4;
    }
  int joff[2]; // column j = 16 (2 wn + j) + 4 (li & 3) -> (kh, kw, c): j = kh * 32 + kw * 4 + c
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jj = (wn * 2 + j) * 16 + 4 * (li & 3);
    joff[j] = ((jj / 32) * 84 + (jj % 32) / 4) * 4 + jj % 4;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; // db1 partials: channels fg * 8 .. + 7 of this lane's pixels

  // zero the tail rows [400, 416) of the dz1 tile once (never written afterwards)
  for (int e = tid; e < 16 * F21_DYS; e += 512)
    sdy[400 * F21_DYS + e] = (bf16)0.f;

  // ---------------- prefetch registers: dz2 (648 vectors), packed observation (1764 vectors), gates (4 atoms)
  constexpr int NXV = 4, NDV = 2;
  struct Regs {
    u32x4 x[NXV], d[NDV], g[APW];
  };
  Regs R;
  auto gload = [&](long n) { // unconditional (clamped) loads: the outstanding count stays static
    n = min(n, P.ns - 1);
    const long nn = n + P.map.n0;
    const long off = (nn / P.map.TP) * P.map.s1 + (nn % P.map.TP) * P.map.s0 + P.map.base; // u32 pixels
    const u32x4 *xs = reinterpret_cast<const u32x4 *>(P.obs + off * 4);
#pragma unroll
    for (int i = 0; i < NXV; ++i)
      R.x[i] = xs[min(tid + 512 * i, 1763)];
    const u32x4 *ds = reinterpret_cast<const u32x4 *>(P.dz2 + n * (long)(81 * 64));
#pragma unroll
    for (int i = 0; i < NDV; ++i)
      R.d[i] = ds[min(tid + 512 * i, 647)];
#pragma unroll
    for (int ak = 0; ak < APW; ++ak)
      R.g[ak] = *reinterpret_cast<const u32x4 *>(P.a1 + (n * 400 + p1[ak]) * (long)32 + fg * 8);
  };
  auto swrite = [&]() {
    auto pk = [](uint32_t lo, uint32_t hi) { return pack_u8_pair_bf16(lo, hi); };
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const int v = tid + 512 * i;
      if (v < 1764) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const uint32_t w0 = R.x[i][2 * d], w1 = R.x[i][2 * d + 1];
          reinterpret_cast<u32x4 *>(sx)[2 * v + d] =
              u32x4{pk(w0 & 255u, (w0 >> 8) & 255u), pk((w0 >> 16) & 255u, w0 >> 24), pk(w1 & 255u, (w1 >> 8) & 255u),
                    pk((w1 >> 16) & 255u, w1 >> 24)};
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NDV; ++i) {
      const int v = tid + 512 * i;
      if (v < 648) { // vector v = 8 channels of dz2 pixel v / 8
        const int pp = v >> 3, row = pp / 9, col = pp - row * 9;
        *reinterpret_cast<u32x4 *>(sd2 + row * RP + col * CP + (v & 7) * 8) = R.d[i];
      }
    }
  };

  typedef __attribute__((address_space(3))) bf16x4 *lds4;
  long n = blockIdx.x;
  gload(n); // (clamped: a workgroup beyond the sample count loads sample ns - 1 and never uses it)
  for (; n < P.ns; n += gridDim.x) {
    swrite();
    u32x4 gate[APW];
#pragma unroll
    for (int ak = 0; ak < APW; ++ak)
      gate[ak] = R.g[ak];
    gload(n + gridDim.x); // next sample: in flight during both MFMA phases
    __syncthreads();
    // ================= phase A: dz1 = gate * (W2d (*) dz2), this wave's atoms of its parity class
#pragma unroll
    for (int ak = 0; ak < APW; ++ak) {
      if (pl + ak * 2 < NATOM) { // wave-uniform
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1v = {0.f, 0.f, 0.f, 0.f};
        const int y = ya[ak], x = xa[ak];
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks) {
          const int tap = ks >> 1, dy = tap >> 1, dx = tap & 1;
          const int sy = y - dy, sxx = x - dx;
          const bool ok = qok[ak] && sy >= 0 && sxx >= 0; // (sy <= 8, sxx <= 8 hold: y, x <= 9 and the pixel is valid)
          const int cy = min(max(sy, 0), 8), cx = min(max(sxx, 0), 8);
          u32x4 b = *reinterpret_cast<const u32x4 *>(sd2 + cy * RP + cx * CP + (ks & 1) * 32 + fg * 8);
          b = (ok && sy <= 8 && sxx <= 8) ? b : zero16();
          Atom<bf16>::mma(W[0][ks], b, a0);
          Atom<bf16>::mma(W[1][ks], b, a1v);
        }
        auto gated = [](uint32_t w, int h, float v) { // ReLU gate on the stored bf16 activation
          const float a = bf16_bits_to_f32(h ? (w >> 16) : (w & 0xFFFFu));
          return a > 0.f ? v : 0.f;
        };
        const u32x4 gt = gate[ak];
        const u32x2 lo = pack4_bf16(gated(gt[0], 0, a0[0]), gated(gt[0], 1, a0[1]), gated(gt[1], 0, a0[2]),
                                    gated(gt[1], 1, a0[3]));
        const u32x2 hi = pack4_bf16(gated(gt[2], 0, a1v[0]), gated(gt[2], 1, a1v[1]), gated(gt[3], 0, a1v[2]),
                                    gated(gt[3], 1, a1v[3]));
        if (qok[ak]) {
          *reinterpret_cast<u32x4 *>(sdy + p1[ak] * F21_DYS + fg * 8) = u32x4{lo[0], lo[1], hi[0], hi[1]};
          // bias gradient: the ROUNDED values the weight-gradient MFMAs multiply with
          bsum[0] += bf16_bits_to_f32(lo[0] & 0xFFFFu);
          bsum[1] += bf16_bits_to_f32(lo[0] >> 16);
          bsum[2] += bf16_bits_to_f32(lo[1] & 0xFFFFu);
          bsum[3] += bf16_bits_to_f32(lo[1] >> 16);
          bsum[4] += bf16_bits_to_f32(hi[0] & 0xFFFFu);
          bsum[5] += bf16_bits_to_f32(hi[0] >> 16);
          bsum[6] += bf16_bits_to_f32(hi[1] & 0xFFFFu);
          bsum[7] += bf16_bits_to_f32(hi[1] >> 16);
        }
      }
    }
    __syncthreads();
    // ================= phase B: dW1 += dz1^T im2col(x)
#pragma unroll
    for (int ks = 0; ks < F21_KS; ++ks) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = KFrag<bf16>::read(sdy, F21_DYS, ks * 32, i * 16, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x2 lo =
            __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(sx + pixoff[ks][0] + joff[j])));
        const u32x2 hi =
            __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(sx + pixoff[ks][1] + joff[j])));
        fb[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          Atom<bf16>::mma(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads(); // the next sample's staging overwrites x / dz2 / (phase A) the dz1 tile
  }
  // ---------------- one slab per workgroup (a workgroup without samples writes zeros)
  float *ow = P.slab_w + (long)blockIdx.x * 32 * 256;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = i * 16 + lg * 4 + r;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        ow[(long)m * 256 + (wn * 2 + j) * 16 + li] = acc[i][j][r] * P.scale;
    }
  // bias: lanes with equal fg hold the same 8 channels; ordered LDS reduction (deterministic)
  __syncthreads();
  float *red = reinterpret_cast<float *>(smem);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    red[tid * 8 + e] = bsum[e];
  __syncthreads();
  if (tid < 32) {
    const int g = tid >> 3, e = tid & 7; // channel tid = g * 8 + e
    float s = 0.f;
    for (int t = 0; t < 512; ++t)
      if (((t & 63) >> 4) == g)
        s += red[t * 8 + e];
    P.slab_b[(long)blockIdx.x * 32 + tid] = s;
  }
}

} // namespace aleppo
