// gemm_pipe.hpp - pipelined bf16 NT GEMM for the fc layer of NetworkImpl (reference src/bin/train.cc:243-246,
// `sequential->push_back(torch::nn::Linear(3136, hidden_size))`) at training batch sizes.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        A: [M][K] bf16 (activations / dh), B: [N][K] bf16 (weights)
//
// The small-tile kernels in gemm.hpp stall once per 64-deep k-step on the `vmcnt(0)` their barrier emits
// while the next tile's LDS-DMA is still in flight: with 2 workgroups per CU they sit at ~11 % of the MFMA
// roof (PMC: 80 % of wave cycles parked in s_waitcnt).  This kernel keeps everything in flight ACROSS
// barriers instead:
//   * 128 x 128 x 64 tiles, 512 threads = 8 waves (2 along m x 4 along n, 64 x 32 outputs per wave);
//   * a ring of NST LDS stage buffers (32 KB each) filled by global_load_lds_dwordx4.  Every batch a wave
//     issues is exactly 4 vector-memory instructions (a k-stage, the gate tile, an epilogue store pass), so
//     "stage s has landed" is the COUNTED `s_waitcnt vmcnt(4 * younger batches)` - never vmcnt(0) inside the
//     loop - followed by a raw s_barrier;
//   * the MFMA fragments are double-buffered in registers: while the 16 MFMAs of stage g run, the 12
//     ds_read_b128 of stage g + 1 and the 4 DMA pieces of stage g + NST are issued between them, so neither
//     the LDS latency nor the DMA issue cost sits between a barrier and the first MFMA;
//   * persistent workgroups: the ring keeps streaming across tile boundaries (job j+1's first stages land
//     while job j's epilogue runs);
//   * LDS rows are 128 B of k, bank conflicts avoided by an XOR swizzle applied on the SOURCE address of
//     the DMA (slot s of row r holds k-chunk s ^ (r & 7)) and again on the fragment reads;
//   * the MFMA operands are swapped (weights as the "A" operand) so that a lane's 4 accumulator registers
//     are 4 CONSECUTIVE n of one row m: 16-byte fp32 / 8-byte bf16 pieces for the epilogue.
//
// MODE 0 (fc forward): split-K partial sums, fp32 [splits][M][N], slice 0 adds the bias.  fc has no
//        activation (train.cc:246 feeds the Linear output straight to the heads), so the consumer
//        (head_train_kernel) just adds the partials.
// MODE 1 (fc dgrad): bf16 [M][N] gated by the ReLU of the forward activation `gate` (same shape).  The gate
//        tile is DMA'd into LDS like a stage (an ordinary global load would make hipcc drain the DMA queue),
//        gated in place and written back with whole 256-byte rows.  Needs >= NST k-stages per job.
//
// (A transposed-gather variant for the weight gradient - both operands k-major, ds_read_b64_tr_b16 fragments - was built
// and measured in rounds 1-2: 51-57 us against 45 us for gemm_tn_kernel; deleted in round 3.)
//
// Wait-count rule used throughout: `s_waitcnt vmcnt(N)` is safe iff N <= the number of vector-memory
// instructions this wave ISSUED after the batch it needs (they retire in issue order).  Under-counting only
// waits longer; epilogue stores are therefore counted only when no wave-instruction can be fully masked off.
#pragma once
#include "gemm.hpp"

namespace aleppo {

struct PipeParams {
  const bf16 *A;
  long lda;
  const bf16 *B;
  long ldb;
  int M, N;
  int tiles_m, tiles_n;
  int nstages; // K / 64
  int splits;  // split-K slices (MODE 0; 1 for MODE 1)
  int njobs;   // tiles_m * tiles_n * splits
  float *out_f32;    // MODE 0: [splits][M][N]
  const float *bias; // MODE 0: [N]
  bf16 *out_bf16;    // MODE 1: [M][N]
  const bf16 *gate;  // MODE 1: [M][N], result = gate > 0 ? C : 0
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `batches` 4-instruction batches of this wave are outstanding (wave-uniform argument)
__device__ __forceinline__ void wait_batches(int batches) {
  switch (batches) {
  case 0: wait_vm<0>(); break;
  case 1: wait_vm<4>(); break;
  case 2: wait_vm<8>(); break;
  case 3: wait_vm<12>(); break;
  case 4: wait_vm<16>(); break;
  case 5: wait_vm<20>(); break;
  case 6: wait_vm<24>(); break;
  default: wait_vm<28>(); break;
  }
}

constexpr int PIPE_STAGE_CHUNKS = 2048; // 16-byte chunks per stage: A rows 0..127 x 8, then B rows 0..127 x 8
template <int MODE, int NST> constexpr size_t gemm_pipe_smem() {
  return (size_t)(NST * PIPE_STAGE_CHUNKS + (MODE == 1 ? 128 * 16 : 0)) * 16;
}

struct PipeFrags {
  u32x4 a[2][4], b[2][2]; // [k-chunk of 32][16-row atom]
};

template <int MODE, int NST> __global__ __launch_bounds__(512) void gemm_pipe_kernel(PipeParams P) {
  static_assert(NST >= 3, "ring depth");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32x4 *ring = reinterpret_cast<u32x4 *>(smem);
  u32x4 *etile = ring + NST * PIPE_STAGE_CHUNKS; // MODE 1: [128 rows][16 chunks], slot s of row r = chunk s ^ (r & 15)
  typedef __attribute__((address_space(1))) const void *gptr;
  typedef __attribute__((address_space(3))) void *lptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int slot = tid & 7, r0 = tid >> 3;    // stage DMA: this thread moves slot `slot` of rows r0 and r0 + 64
  const int eslot = tid & 15, er0 = tid >> 4; // gate / output tile: slot eslot of rows er0 + 32 i

  // Job -> workgroup map.  Workgroup b runs on XCD b % 8 (round-robin dispatch; a speed assumption only); jobs are
  // grouped so that the operand panels an XCD touches at any time stay inside its 4 MB L2.
  // Two XCD-grouped maps: (x) m-tiles divide by 8 -> every XCD owns tiles_m / 8 row panels and its workgroups sweep the
  // n-tiles together; (p) otherwise the (n-tile, slice) pairs are dealt round-robin to the XCDs and the m-tiles of a
  // pair run side by side on one XCD, sharing that pair's B stream (the wgrad: 4 m-tiles re-use each a3 tile).
  const bool grouped = gridDim.x % 8 == 0;
  const bool xmap = grouped && P.tiles_m % 8 == 0, pmap = grouped && !xmap;
  const int xcd = (int)blockIdx.x % 8;
  const int tmx = xmap ? P.tiles_m / 8 : P.tiles_m; // m-tiles per job group
  const int npairs = P.tiles_n * P.splits;
  const int pairs_g = pmap ? (npairs - xcd + 7) / 8 : npairs;          // (n-tile, slice) pairs of this group
  const int jobs_g = tmx * pairs_g;                                     // jobs of this group
  const int wg_l = grouped ? (int)blockIdx.x / 8 : (int)blockIdx.x;     // this workgroup within its group
  const int nwg_l = grouped ? (int)gridDim.x / 8 : (int)gridDim.x;
  const int tm_base = xmap ? xcd * tmx : 0;
  const int nj = (jobs_g - wg_l + nwg_l - 1) / nwg_l; // jobs of this workgroup
  if (nj <= 0)
    return;
  struct Job {
    int m0, n0, slice, ks0, nks;
  };
  auto decode = [&](int ord) {
    const int j = wg_l + ord * nwg_l;
    int slice, tm, tn;
    if (pmap) { // m-tile fastest, then this group's pairs
      const int pr = xcd + 8 * (j / tmx);
      tm = j % tmx;
      slice = pr % P.splits;
      tn = pr / P.splits;
    } else {
      slice = j % P.splits;
      const int t = j / P.splits;
      tm = tm_base + t % tmx;
      tn = t / tmx;
    }
    const int ks0 = (int)((long)slice * P.nstages / P.splits), ks1 = (int)((long)(slice + 1) * P.nstages / P.splits);
    return Job{tm * 128, tn * 128, slice, ks0, ks1 - ks0};
  };
  int G = 0; // total stages of this workgroup
#pragma nounroll
  for (int o = 0; o < nj; ++o)
    G += decode(o).nks;

  // ---------------- producer (DMA) cursor: stage s is issued NST stages ahead of its consumption
  int pord = 0, pk = 0, pissued = 0, pbuf = 0;
  Job pjob = decode(0);
  const bf16 *pa[2], *pb[2];
  auto set_src = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + 64 * i, sw = (slot ^ (r & 7)) * 8;
      pa[i] = P.A + (long)min(pjob.m0 + r, P.M - 1) * P.lda + (long)pjob.ks0 * 64 + sw;
      pb[i] = P.B + (long)min(pjob.n0 + r, P.N - 1) * P.ldb + (long)pjob.ks0 * 64 + sw;
    }
  };
  set_src();
  // one batch = 4 LDS-DMA instructions per wave (pieces 0,1: A rows r0, r0 + 64; 2,3: B rows), then advance()
  auto piece = [&](int q) {
    u32x4 *dst = ring + pbuf * PIPE_STAGE_CHUNKS + wave * 64;
    const long ka = (long)pk * 64, kb = (long)pk * 64;
    if (q < 2)
      __builtin_amdgcn_global_load_lds((gptr)(pa[q] + ka), (lptr)(dst + 512 * q), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((gptr)(pb[q - 2] + kb), (lptr)(dst + 1024 + 512 * (q - 2)), 16, 0, 0);
  };
  auto advance = [&]() {
    pbuf = pbuf + 1 == NST ? 0 : pbuf + 1;
    ++pissued;
    if (++pk == pjob.nks && ++pord < nj) {
      pjob = decode(pord);
      pk = 0;
      set_src();
    }
  };

  // fragment reads of the stage in ring buffer `buf`
  auto read_a = [&](PipeFrags &f, int buf, int kc, int i) {
    const int r = wm * 64 + i * 16 + fr;
    f.a[kc][i] = ring[buf * PIPE_STAGE_CHUNKS + r * 8 + ((kc * 4 + fg) ^ (r & 7))];
  };
  auto read_b = [&](PipeFrags &f, int buf, int kc, int j) {
    const int r = wn * 32 + j * 16 + fr;
    f.b[kc][j] = ring[buf * PIPE_STAGE_CHUNKS + 1024 + r * 8 + ((kc * 4 + fg) ^ (r & 7))];
  };

  f32x4 acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

#pragma nounroll
  for (int i = 0; i < NST && pissued < G; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      piece(q);
    advance();
  }

  int cord = 0, ck = 0, cbuf = 0; // cbuf: ring buffer of the stage whose fragments are being READ (stage g + 1)
  Job cjob = decode(0);
  int ext[NST - 1]; // extra batches (gate tile, epilogue stores) issued in the previous NST-1 iterations, newest first
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    ext[i] = 0;

  PipeFrags F0, F1;
  { // stage 0: wait, barrier, read its fragments
    wait_batches(min(NST - 1, G - 1));
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        read_b(F0, 0, kc, j);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        read_a(F0, 0, kc, i);
    }
    cbuf = 1;
  }

  // one stage: MFMAs on `cur` (stage g) while `nxt` (stage g + 1) is read and stage g + NST is issued
  auto stage = [&](int g, PipeFrags &cur, PipeFrags &nxt) {
    const bool has_next = g + 1 < G;
    // this wave's reads of stage g (issued one stage ago) are complete -> after the barrier its buffer may be refilled
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (has_next) { // stage g + 1 has landed once only the batches issued after it are outstanding
      int young = min(NST - 2, G - 2 - g);
#pragma unroll
      for (int i = 0; i < NST - 1; ++i)
        young += ext[i];
      asm volatile("" : "+s"(young)); // opaque: stops hipcc from cloning the loop body per (ext[], young) state
      wait_batches(young);
    }
    __builtin_amdgcn_s_barrier();
    int extra = 0;
    const bool prod = pissued < G; // stage g + NST goes into the buffer stage g occupied
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { // 4 x (4 MFMAs, 3 fragment reads, 1 DMA piece)
#pragma unroll
      for (int u = 0; u < 4; ++u) { // weights as the A operand: acc[i][j][r] = C[m = ..+fr][n = ..+4 fg + r]
        const int kc = q >> 1, i = (q & 1) * 2 + (u >> 1), j = u & 1;
        Atom<bf16>::mma(cur.b[kc][j], cur.a[kc][i], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) {
        const int kc = q >> 1, h = q & 1;
        read_b(nxt, cbuf, kc, h);
        read_a(nxt, cbuf, kc, 2 * h);
        read_a(nxt, cbuf, kc, 2 * h + 1);
      }
      if (prod)
        piece(q);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_next)
      cbuf = cbuf + 1 == NST ? 0 : cbuf + 1;
    if (prod)
      advance();
    if constexpr (MODE == 1) {
      if (ck == 0) { // gate tile of this job (its buffer was last read before the barrier above)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = er0 + 32 * i;
          const bf16 *src =
              P.gate + (long)min(cjob.m0 + r, P.M - 1) * P.N + min(cjob.n0 + ((eslot ^ (r & 15)) * 8), P.N - 8);
          __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(etile + wave * 64 + 512 * i), 16, 0, 0);
        }
        extra += 1;
      }
    }
    if (++ck == cjob.nks) { // ---------------- epilogue of this job
      const int mrow = cjob.m0 + wm * 64 + fr, ncol = cjob.n0 + wn * 32 + fg * 4;
      if constexpr (MODE == 0) {
        float *out = P.out_f32 + (long)cjob.slice * P.M * P.N;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = ncol + j * 16;
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (MODE == 0 && cjob.slice == 0 && n < P.N)
            b = *reinterpret_cast<const f32x4 *>(P.bias + n);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = mrow + i * 16;
            if (m < P.M && n < P.N)
              *reinterpret_cast<f32x4 *>(out + (long)m * P.N + n) = acc[i][j] + b;
          }
        }
        // 8 stores per wave = 2 batches - counted only when no lane can be masked off (see the rule on top)
        extra += (cjob.m0 + 128 <= P.M && cjob.n0 + 128 <= P.N) ? 2 : 0;
      } else {
        // The gate batch is older than every stage batch still allowed in flight here (stages g + 2 .. g + NST were
        // issued after it because a job has at least NST k-stages), so waiting for all but those retires it.
        wait_batches(max(0, min(NST - 1, G - 2 - g)));
        __builtin_amdgcn_s_barrier();
        uint8_t *eb = reinterpret_cast<uint8_t *>(etile);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int r = wm * 64 + i * 16 + fr, c = wn * 32 + j * 16 + fg * 4; // 4 columns = 8 bytes of chunk c / 8
            u32x2 *p = reinterpret_cast<u32x2 *>(eb + r * 256 + (((c >> 3) ^ (r & 15)) * 16) + (c & 7) * 2);
            const u32x2 gt = *p;
            auto on = [](uint32_t w, int hi) { // bf16 > 0: magnitude bits set, sign clear
              const uint32_t h = hi ? (w >> 16) : (w & 0xFFFFu);
              return (h & 0x7FFFu) != 0 && (h & 0x8000u) == 0;
            };
            const float v0 = on(gt[0], 0) ? acc[i][j][0] : 0.f, v1 = on(gt[0], 1) ? acc[i][j][1] : 0.f;
            const float v2 = on(gt[1], 0) ? acc[i][j][2] : 0.f, v3 = on(gt[1], 1) ? acc[i][j][3] : 0.f;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            const bf16x2 lo = {(bf16)v0, (bf16)v1}, hi = {(bf16)v2, (bf16)v3};
            *p = u32x2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) { // whole 256-byte rows: 16 lanes x 16 B
          const int r = er0 + 32 * i, m = cjob.m0 + r, n = cjob.n0 + ((eslot ^ (r & 15)) * 8);
          const u32x4 v = etile[r * 16 + eslot];
          if (m < P.M && n < P.N)
            *reinterpret_cast<u32x4 *>(P.out_bf16 + (long)m * P.N + n) = v;
        }
        extra += (cjob.m0 + 128 <= P.M) ? 1 : 0; // every wave-instruction has live lanes iff all 128 rows exist
      }
      zero_acc();
      ck = 0;
      if (++cord < nj)
        cjob = decode(cord);
    }
    asm volatile("" : "+s"(extra));
#pragma unroll
    for (int i = NST - 2; i > 0; --i)
      ext[i] = ext[i - 1];
    ext[0] = extra;
  };

#pragma nounroll
  for (int g = 0; g < G; g += 2) { // (without nounroll hipcc expands this loop into ~90k lines of ISA)
    stage(g, F0, F1);
    if (g + 1 < G)
      stage(g + 1, F1, F0);
  }
}

} // namespace aleppo
