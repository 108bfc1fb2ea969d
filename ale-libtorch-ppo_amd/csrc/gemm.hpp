// gemm.hpp - MFMA gather-GEMM kernels for the Nature-CNN stack on gfx950 (CDNA4).
//
// Two kernel templates cover every conv / linear forward, dgrad and wgrad of
// NetworkImpl (reference src/bin/train.cc:230-265) as IMPLICIT GEMMs - im2col is never
// materialised in HBM; the loaders below generate the gather addresses and the tiles are
// staged through LDS:
//
//   gemm_nt : C[m,n] = epi( sum_k A(m,k) * B(n,k) )      A,B "k-contiguous"   (fwd, dgrad)
//   gemm_tn : S[z][m,n] = sum_{p in split z} A(p,m) * B(p,n)                  (wgrad, split-K slabs)
//
// Arithmetic type T is float (exact-parity path: v_mfma_f32_16x16x4_f32, an fp32 fma chain) or
// __bf16 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).  Both atoms consume "16 bytes of
// consecutive k per lane", so one tile geometry serves both: a stage is 128 bytes of k per row
// (KT = 32 floats or 64 bf16), i.e. two 64-byte chunks; lane l of a wave holds bytes
// [16*(l>>4), +16) of the chunk for row/col (l&15).  For float the 4 lanes-groups x 4 floats
// are a permutation of the chunk's 16 k's, identical for A and B, so the sum is unchanged.
//
// Wave = 64 lanes, workgroup = 256 threads = 4 waves arranged WM x WN over the BM x BN tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aleppo {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __bf16 bf16;

__device__ __forceinline__ u32x4 zero16() { return u32x4{0u, 0u, 0u, 0u}; }

template <class T> struct Atom;
template <> struct Atom<float> {
  static constexpr int VE = 4;  // elements per 16-byte vector
  static constexpr int KT = 32; // elements of k per stage (128 B)
  static __device__ __forceinline__ void mma(const u32x4 &a, const u32x4 &b, f32x4 &c) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to(float v) { return v; }
};
template <> struct Atom<bf16> {
  static constexpr int VE = 8;
  static constexpr int KT = 64;
  static __device__ __forceinline__ void mma(const u32x4 &a, const u32x4 &b, f32x4 &c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                0);
  }
  static __device__ __forceinline__ bf16 to(float v) { return (bf16)v; }
};

// two small integers (0..255: exact in bf16's 8 significand bits) -> packed bf16 pair {lo, hi}: the upper halves of
// the two f32 encodings, gathered by ONE v_perm_b32 (3 VALU per pair with the v_cvt_f32_ubyteN the casts become,
// instead of 4 with shift + and_or: the u8 -> bf16 widening is ~40 % of the conv1 kernels' VALU work)
__device__ __forceinline__ uint32_t pack_u8_pair_bf16(uint32_t lo, uint32_t hi) {
  return __builtin_amdgcn_perm(__float_as_uint((float)hi), __float_as_uint((float)lo), 0x07060302u);
}

// uint8 -> T widening of one 16-byte LDS vector worth of elements (exact: 0..255 fit bf16's 8 bits)
template <class T> __device__ __forceinline__ u32x4 widen_u8(const uint8_t *p);
template <> __device__ __forceinline__ u32x4 widen_u8<float>(const uint8_t *p) {
  const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
  f32x4 f = {(float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24)};
  return __builtin_bit_cast(u32x4, f);
}
template <> __device__ __forceinline__ u32x4 widen_u8<bf16>(const uint8_t *p) {
  const u32x2 w = *reinterpret_cast<const u32x2 *>(p);
  auto pk = [](uint32_t lo, uint32_t hi) { // two exact small ints -> packed bf16 pair
    return pack_u8_pair_bf16(lo, hi);
  };
  return u32x4{pk(w[0] & 255u, (w[0] >> 8) & 255u), pk((w[0] >> 16) & 255u, w[0] >> 24),
               pk(w[1] & 255u, (w[1] >> 8) & 255u), pk((w[1] >> 16) & 255u, w[1] >> 24)};
}

// ------------------------------------------------------------------------------------------------
// Loaders.  row(m) precomputes per-row state once; load(row, k) returns the 16-byte vector holding
// elements [k, k+VE) of that row (zeros outside the matrix).  set_z() selects a grid.z slice.
// ------------------------------------------------------------------------------------------------

// Dense row-major [rows][ld] matrix of T.
template <class T> struct DenseLoader {
  struct P {
    const T *a;
    long ld;
    long zstride;
  };
  struct Row {
    long off;
    bool ok;
  };
  static __device__ __forceinline__ void set_z(P &p, int z) { p.a += (long)z * p.zstride; }
  static __device__ __forceinline__ Row row(const P &p, int m, int M) { return Row{(long)m * p.ld, m < M}; }
  static __device__ __forceinline__ u32x4 load(const P &p, const Row &r, int k, int K) {
    if (!r.ok || k >= K)
      return zero16();
    return *reinterpret_cast<const u32x4 *>(p.a + r.off + k);
  }
};

// Implicit im2col of an NHWC tensor (C innermost): row m = (sample n, oy, ox), k = (kh, kw*C + c).
// One kh segment (KW*C elements) is contiguous in memory.  Sample n lives at
//   x + (n / TP) * s1 + (n % TP) * s0 + base      (rollout buffer slots are [E][T+1])
// InT = uint8_t reads the packed 4-frame stack (conv1) and widens to T.
template <class T, class InT, int PIX, int OW, int STRIDE, int IW, int C, int KW> struct ConvGatherLoader {
  static constexpr int SEG = KW * C, PITCH = IW * C;
  struct P {
    const InT *x;
    int TP;
    long s1, s0, base;
    int n0;
  };
  struct Row {
    long off;
    bool ok;
  };
  static __device__ __forceinline__ void set_z(P &, int) {}
  static __device__ __forceinline__ Row row(const P &p, int m, int M) {
    if (m >= M)
      return Row{0, false};
    const int nl = m / PIX, pix = m - nl * PIX, oy = pix / OW, ox = pix - oy * OW;
    const int n = nl + p.n0, q = n / p.TP, r = n - q * p.TP;
    return Row{(long)q * p.s1 + (long)r * p.s0 + p.base + (long)((oy * STRIDE) * IW + ox * STRIDE) * C, true};
  }
  static __device__ __forceinline__ u32x4 load(const P &p, const Row &r, int k, int K) {
    if (!r.ok || k >= K)
      return zero16();
    const int kq = k / SEG, kr = k - kq * SEG;
    const InT *ptr = p.x + r.off + kq * PITCH + kr;
    if constexpr (sizeof(InT) == 1)
      return widen_u8<T>(reinterpret_cast<const uint8_t *>(ptr));
    else
      return *reinterpret_cast<const u32x4 *>(ptr);
  }
};

// Transposed-conv gather for dgrad: row m = (n, y, x) on a PH x PW grid, k = (tap, oc),
// tap = (dy, dx) on a TH x TW grid; source pixel (y - dy, x - dx) of dY [n][OH][OW][OC], zero
// outside.  conv3 (3x3, s1): grid 9x9, taps 3x3.  conv2 (4x4, s2): one launch slice per input
// parity class, grid 10x10, taps 2x2 (DESIGN.md "dgrad by parity class").
template <class T, int PH, int PW, int OH, int OW, int OC, int TW> struct DgradGatherLoader {
  struct P {
    const T *dy;
  };
  struct Row {
    long base;
    int y, x;
    bool ok;
  };
  static __device__ __forceinline__ void set_z(P &, int) {}
  static __device__ __forceinline__ Row row(const P &, int m, int M) {
    if (m >= M)
      return Row{0, 0, 0, false};
    const int n = m / (PH * PW), rem = m - n * (PH * PW), y = rem / PW, x = rem - y * PW;
    return Row{(long)n * (OH * OW * OC), y, x, true};
  }
  static __device__ __forceinline__ u32x4 load(const P &p, const Row &r, int k, int K) {
    if (!r.ok || k >= K)
      return zero16();
    const int tap = k / OC, oc = k - tap * OC, dy = tap / TW, dx = tap - dy * TW;
    const int sy = r.y - dy, sx = r.x - dx;
    if (sy < 0 || sy >= OH || sx < 0 || sx >= OW)
      return zero16();
    return *reinterpret_cast<const u32x4 *>(p.dy + r.base + (long)(sy * OW + sx) * OC + oc);
  }
};

// ------------------------------------------------------------------------------------------------
// Epilogues.  row(m) once per accumulator row, store(row, n, v) per element.
// ------------------------------------------------------------------------------------------------

// out[m*ld + n] = act( v*scale + bias[n] )
template <class OutT, bool RELU> struct EpiBiasAct {
  struct P {
    OutT *out;
    const float *bias;
    long ld;
    float scale;
  };
  struct Row {
    long off;
    bool ok;
  };
  static __device__ __forceinline__ void set_z(P &, int) {}
  static __device__ __forceinline__ Row row(const P &p, int m, int M) { return Row{(long)m * p.ld, m < M}; }
  static __device__ __forceinline__ void store(const P &p, const Row &r, int n, int N, float v) {
    if (!r.ok || n >= N)
      return;
    v = v * p.scale + p.bias[n];
    if (RELU)
      v = v > 0.f ? v : 0.f;
    p.out[r.off + n] = (OutT)v;
  }
};

// split-K partial sums: out[z][m*ld + n] = v  (no bias; the consumer adds the slices in fixed order)
struct EpiPartial {
  struct P {
    float *out;
    long ld;
    long zstride;
  };
  struct Row {
    long off;
    bool ok;
  };
  static __device__ __forceinline__ void set_z(P &p, int z) { p.out += (long)z * p.zstride; }
  static __device__ __forceinline__ Row row(const P &p, int m, int M) { return Row{(long)m * p.ld, m < M}; }
  static __device__ __forceinline__ void store(const P &p, const Row &r, int n, int N, float v) {
    if (r.ok && n < N)
      p.out[r.off + n] = v;
  }
};

// dgrad epilogue: out = v * [act > 0]  (gradient through the ReLU that produced `act`).
// MAP 0: linear rows (off = m*ld).  MAP 1: conv2 parity class z=(py,px): row (n,y',x') ->
// pixel (2y'+py, 2x'+px) of the [n][20][20][32] tensor.
template <class T, int MAP> struct EpiReluMask {
  struct P {
    T *out;
    const T *act;
    long ld;
    int z;
  };
  struct Row {
    long off;
    bool ok;
  };
  static __device__ __forceinline__ void set_z(P &p, int z) { p.z = z; }
  static __device__ __forceinline__ Row row(const P &p, int m, int M) {
    if (m >= M)
      return Row{0, false};
    if constexpr (MAP == 0)
      return Row{(long)m * p.ld, true};
    else {
      const int n = m / 100, rem = m - n * 100, y = rem / 10, x = rem - y * 10;
      const int py = p.z >> 1, px = p.z & 1;
      return Row{((long)(n * 20 + 2 * y + py) * 20 + 2 * x + px) * 32, true};
    }
  }
  static __device__ __forceinline__ void store(const P &p, const Row &r, int n, int N, float v) {
    if (!r.ok || n >= N)
      return;
    const float a = (float)p.act[r.off + n];
    p.out[r.off + n] = (T)(a > 0.f ? v : 0.f);
  }
};

// ------------------------------------------------------------------------------------------------
// gemm_nt: C = epi(A * B^T).  LDS rows are 128 B of k + 16 B pad (conflict-free ds_read_b128 for
// the 16-row x 16-B fragment pattern), double-buffered; next stage is prefetched into registers
// while the current one feeds the MFMAs; one barrier per stage.
// ------------------------------------------------------------------------------------------------
constexpr int LDS_ROW_V = 9; // 144 B per row, in 16-byte vectors

template <class T, class AL, class BL, class EP, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(typename AL::P ap, typename BL::P bp, typename EP::P ep, int M,
                                                       int N, int K) {
  using AT = Atom<T>;
  constexpr int VE = AT::VE, KT = AT::KT;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int AV = BM / 32, BV = BN / 32; // 16-byte vectors staged per thread per stage
  static_assert(WM * WN == 4 && BM % 32 == 0 && BN % 32 == 0 && WTM % 16 == 0 && WTN % 16 == 0, "tile");
  __shared__ u32x4 sA[2][BM * LDS_ROW_V];
  __shared__ u32x4 sB[2][BN * LDS_ROW_V];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  AL::set_z(ap, blockIdx.z);
  BL::set_z(bp, blockIdx.z);
  EP::set_z(ep, blockIdx.z);

  const int kv = tid & 7, r0 = tid >> 3; // this thread stages vector kv of rows r0, r0+32, ...
  typename AL::Row arow[AV];
  typename BL::Row brow[BV];
#pragma unroll
  for (int i = 0; i < AV; ++i)
    arow[i] = AL::row(ap, m0 + r0 + 32 * i, M);
#pragma unroll
  for (int i = 0; i < BV; ++i)
    brow[i] = BL::row(bp, n0 + r0 + 32 * i, N);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  // register prefetch TWO stages ahead (two register sets): a stage's global loads have a full
  // stage of MFMAs plus the LDS phase of the previous one to land before they are written to LDS
  u32x4 ra[2][AV], rb[2][BV];
  const int nk = (K + KT - 1) / KT;
  auto gload = [&](int ks, int set) {
    const int k = ks * KT + kv * VE;
#pragma unroll
    for (int i = 0; i < AV; ++i)
      ra[set][i] = AL::load(ap, arow[i], k, K);
#pragma unroll
    for (int i = 0; i < BV; ++i)
      rb[set][i] = BL::load(bp, brow[i], k, K);
  };
  auto swrite = [&](int buf, int set) {
#pragma unroll
    for (int i = 0; i < AV; ++i)
      sA[buf][(r0 + 32 * i) * LDS_ROW_V + kv] = ra[set][i];
#pragma unroll
    for (int i = 0; i < BV; ++i)
      sB[buf][(r0 + 32 * i) * LDS_ROW_V + kv] = rb[set][i];
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      u32x4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = sA[buf][(wm * WTM + i * 16 + fr) * LDS_ROW_V + kc * 4 + fg];
#pragma unroll
      for (int j = 0; j < NI; ++j)
        fb[j] = sB[buf][(wn * WTN + j * 16 + fr) * LDS_ROW_V + kc * 4 + fg];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          AT::mma(fa[i], fb[j], acc[i][j]);
    }
  };

  gload(0, 0);
  if (nk > 1)
    gload(1, 1);
  swrite(0, 0);
  __syncthreads();
  // two stages per trip so that the register-set index is a compile-time constant
  for (int ks = 0; ks < nk; ks += 2) {
    if (ks + 2 < nk)
      gload(ks + 2, 0);
    compute(0);
    if (ks + 1 < nk)
      swrite(1, 1);
    __syncthreads();
    if (ks + 1 < nk) {
      if (ks + 3 < nk)
        gload(ks + 3, 1);
      compute(1);
      if (ks + 2 < nk)
        swrite(0, 0);
      __syncthreads();
    }
  }
  // C/D layout of the 16x16 atoms: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const typename EP::Row er = EP::row(ep, m0 + wm * WTM + i * 16 + fg * 4 + r, M);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        EP::store(ep, er, n0 + wn * WTN + j * 16 + fr, N, acc[i][j][r]);
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt_dma: dense C = epi(A * B^T) with DIRECT-TO-LDS staging (global_load_lds_dwordx4).  The
// register-staged kernel above tops out on the LDS *write* path (ds_write_b128 ~ 80 B/clk/CU: the same
// fc GEMM reaches 64 % of the fp32 MFMA roof but only ~10 % of the bf16 one at identical bytes per
// stage); the DMA writes LDS without VGPRs or ds_write.  Its destination is wave-uniform base + lane*16,
// so LDS rows are the unpadded 128 B of a stage and bank conflicts are avoided by an XOR swizzle applied
// on the SOURCE side: the 16-byte slot p of row r holds k-chunk p ^ (r & 7); fragment reads use the same
// involution.  Two LDS buffers; the next stage's DMA is in flight during this stage's MFMAs; the
// barrier's vmcnt(0) retires it.  Requires K % KT == 0 (rows beyond M / N are clamped and masked by the
// epilogue).
// ------------------------------------------------------------------------------------------------
// CH = 16-byte chunks of k per row and stage (8 -> 128 B, 16 -> 256 B: half the barriers, twice the bytes in
// flight per workgroup).
template <class T, class EP, int BM, int BN, int WM, int WN, int CH = 8>
__global__ __launch_bounds__(256) void gemm_nt_dma_kernel(const T *A, long lda, const T *B, long ldb, typename EP::P ep,
                                                           int M, int N, int K) {
  using AT = Atom<T>;
  constexpr int VE = AT::VE, KT = VE * CH;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int RPP = 256 / CH; // rows covered by one pass of the 256 threads
  constexpr int AV = BM / RPP, BV = BN / RPP;
  static_assert(WM * WN == 4 && BM % RPP == 0 && BN % RPP == 0, "tile");
  __shared__ __attribute__((aligned(16))) u32x4 sA[2][BM * CH];
  __shared__ __attribute__((aligned(16))) u32x4 sB[2][BN * CH];
  typedef __attribute__((address_space(1))) const void *gptr;
  typedef __attribute__((address_space(3))) void *lptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  EP::set_z(ep, blockIdx.z);
  // this thread moves slot (tid % CH) of rows (tid / CH) + RPP i: source chunk = slot ^ (row & 7)
  const int slot = tid % CH, r0 = tid / CH;
  const T *asrc[AV], *bsrc[BV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int r = r0 + RPP * i;
    asrc[i] = A + (long)min(m0 + r, M - 1) * lda + ((slot ^ (r & 7)) * VE);
  }
#pragma unroll
  for (int i = 0; i < BV; ++i) {
    const int r = r0 + RPP * i;
    bsrc[i] = B + (long)min(n0 + r, N - 1) * ldb + ((slot ^ (r & 7)) * VE);
  }
  auto stage = [&](int buf, int ks) {
    const int k = ks * KT;
#pragma unroll
    for (int i = 0; i < AV; ++i)
      __builtin_amdgcn_global_load_lds((gptr)(asrc[i] + k), (lptr)&sA[buf][wave * 64 + 256 * i], 16, 0, 0);
#pragma unroll
    for (int i = 0; i < BV; ++i)
      __builtin_amdgcn_global_load_lds((gptr)(bsrc[i] + k), (lptr)&sB[buf][wave * 64 + 256 * i], 16, 0, 0);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  const int nk = K / KT;
  stage(0, 0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk)
      stage(buf ^ 1, ks + 1);
#pragma unroll
    for (int kc = 0; kc < CH / 4; ++kc) {
      u32x4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * WTM + i * 16 + fr;
        fa[i] = sA[buf][r * CH + ((kc * 4 + fg) ^ (r & 7))];
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int r = wn * WTN + j * 16 + fr;
        fb[j] = sB[buf][r * CH + ((kc * 4 + fg) ^ (r & 7))];
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          AT::mma(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const typename EP::Row er = EP::row(ep, m0 + wm * WTM + i * 16 + fg * 4 + r, M);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        EP::store(ep, er, n0 + wn * WTN + j * 16 + fr, N, acc[i][j][r]);
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: weight gradients.  Reduction index p (sample-pixel) is the OUTER dimension of both
// operands in memory: A(p, m) = dY[p][m] (dense), B(p, n) = im2col(X)[p][n] (gather loader with
// row = p).  A stage is KP = KT pixels; tiles land in LDS as [p][m] / [p][n] with 16-byte vector
// writes and fragments are read k-strided (ds_read_b32 for float, ds_read_b64_tr_b16 for bf16).
// grid.z = split-K slice; each slice writes an fp32 partial slab [z][M][N] that
// reduce_slabs_kernel sums in fixed order (deterministic; no atomics).
// BIAS: n-tile 0 also emits column sums of A (the bias gradient) into bias_slab [z][M].
// out_scale multiplies the weight slab (conv1: the 1/255 input scaling that forward folds into its epilogue).
// ------------------------------------------------------------------------------------------------
// k-strided fragment read from an LDS tile stored [k][m] (m contiguous): returns the 16-byte MFMA operand
// of lane (i = lane&15 -> column m0+i, g = lane>>4 -> this lane's k group) for the atom-k step whose
// first row is krow0.  float: 4 x ds_read_b32 of rows krow0+4g+j.  bf16: 2 x ds_read_b64_tr_b16 (gfx950
// hardware transpose): rows krow0+4g..+3 and krow0+16+4g..+3 - a permutation of the step's 32 k's that
// is identical for both operands, so the product is unchanged; with a row stride = 8 (mod 16) dwords
// the 8 rows a half-wave touches tile the 64 banks exactly (conflict-free).
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
template <class T> struct KFrag;
template <> struct KFrag<float> {
  static constexpr int PAD = 4; // row stride = 4 (mod 8) dwords
  static __device__ __forceinline__ u32x4 read(const float *tile, int stride, int krow0, int m0, int lane) {
    const float *b = tile + (krow0 + (lane >> 4) * 4) * stride + m0 + (lane & 15);
    f32x4 f = {b[0], b[stride], b[2 * stride], b[3 * stride]};
    return __builtin_bit_cast(u32x4, f);
  }
};
template <> struct KFrag<bf16> {
  static constexpr int PAD = 16; // elements; (BM/2 + 8) dwords = 8 (mod 16) for BM in {32,64,128}
  static __device__ __forceinline__ u32x4 read(const bf16 *tile, int stride, int krow0, int m0, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const bf16 *p = tile + (krow0 + 4 * g + (i >> 2)) * stride + m0 + 4 * (i & 3);
    typedef __attribute__((address_space(3))) bf16x4 *lds4;
    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)p));
    const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(p + 16 * stride)));
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  }
};

template <class T, class AL, class BL, int BM, int BN, int WM, int WN, bool BIAS>
__global__ __launch_bounds__(256) void gemm_tn_kernel(typename AL::P ap, typename BL::P bp, float *slab,
                                                       float *bias_slab, int M, int N, int Ktot, int kchunk,
                                                       float out_scale, int xcd_swizzle) {
  using AT = Atom<T>;
  constexpr int VE = AT::VE, KP = AT::KT; // pixels per stage
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NI = WTN / 16;
  constexpr int TPR = 256 / KP;           // threads per pixel row (4 bf16 / 8 float)
  constexpr int AVR = BM / VE, BVR = BN / VE; // vectors per pixel row
  constexpr int AV = (AVR + TPR - 1) / TPR, BV = (BVR + TPR - 1) / TPR;
  constexpr int SAE = BM + KFrag<T>::PAD, SBE = BN + KFrag<T>::PAD; // LDS row length in elements
  static_assert(WM * WN == 4 && WTM % 16 == 0 && WTN % 16 == 0, "tile");
  __shared__ __attribute__((aligned(16))) T sA[2][KP * SAE];
  __shared__ __attribute__((aligned(16))) T sB[2][KP * SBE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // Workgroups are dealt round-robin to the 8 XCDs in linear block order.  The gridDim.x m-tiles that share one
  // (n-tile, slice) B stream would land on 8 different L2s and each fetch it again (PMC: 211 MB fetched for 30 MB
  // of fc wgrad operands); with xcd_swizzle every XCD takes a CONTIGUOUS chunk of the work list instead, so
  // those m-tiles run side by side on one XCD.  (Needs a block count that is a multiple of 8; launched 1-D.)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (xcd_swizzle) { // 1-D launch (blockIdx.x IS the dispatch order); xcd_swizzle = m-tiles | n-tiles << 12
    const int gx = xcd_swizzle & 4095, gy = xcd_swizzle >> 12;
    const int lin = blockIdx.x, per = gridDim.x / 8;
    const int w = (lin % 8) * per + lin / 8;
    bx = w % gx;
    by = (w / gx) % gy;
    bz = w / (gx * gy);
  }
  const int m0 = bx * BM, n0 = by * BN, z = bz;
  const int kbeg = z * kchunk, kend = min(Ktot, kbeg + kchunk);
  const int pl = tid / TPR, tv = tid % TPR; // this thread stages pixel row pl of every stage

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
      acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  u32x4 ra[AV], rb[BV];
  const int nk = (kend - kbeg + KP - 1) / KP;
  auto gload = [&](int ks) {
    const int p = kbeg + ks * KP + pl;
    const typename AL::Row ar = AL::row(ap, p, kend);
    const typename BL::Row br = BL::row(bp, p, kend);
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tv + i * TPR;
      ra[i] = v < AVR ? AL::load(ap, ar, m0 + v * VE, M) : zero16();
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int v = tv + i * TPR;
      rb[i] = v < BVR ? BL::load(bp, br, n0 + v * VE, N) : zero16();
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tv + i * TPR;
      if (v < AVR)
        *reinterpret_cast<u32x4 *>(&sA[buf][pl * SAE + v * VE]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int v = tv + i * TPR;
      if (v < BVR)
        *reinterpret_cast<u32x4 *>(&sB[buf][pl * SBE + v * VE]) = rb[i];
    }
  };

  if (nk > 0) {
    gload(0);
    swrite(0);
  }
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk)
      gload(ks + 1);
#pragma unroll
    for (int kc = 0; kc < KP / (4 * VE); ++kc) { // one atom-k (16 floats / 32 bf16) per iteration
      u32x4 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = KFrag<T>::read(sA[buf], SAE, kc * 4 * VE, wm * WTM + i * 16, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        fb[j] = KFrag<T>::read(sB[buf], SBE, kc * 4 * VE, wn * WTN + j * 16, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          AT::mma(fa[i], fb[j], acc[i][j]);
    }
    if (BIAS && by == 0 && tid < BM) { // (the LOGICAL n-tile 0: exactly one block per (m-tile, slice))
      float s = 0.f;
#pragma unroll 8
      for (int p = 0; p < KP; ++p)
        s += (float)sA[buf][p * SAE + tid];
      bsum += s;
    }
    if (ks + 1 < nk)
      swrite(buf ^ 1);
    __syncthreads();
  }
  float *out = slab + (long)z * M * N;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * WTM + i * 16 + fg * 4 + r;
      if (m < M) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = n0 + wn * WTN + j * 16 + fr;
          if (n < N)
            out[(long)m * N + n] = acc[i][j][r] * out_scale;
        }
      }
    }
  if (BIAS && by == 0 && tid < BM && m0 + tid < M)
    bias_slab[(long)z * M + m0 + tid] = bsum;
}

} // namespace aleppo
