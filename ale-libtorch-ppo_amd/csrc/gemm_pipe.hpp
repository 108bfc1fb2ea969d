// gemm_pipe.hpp - pipelined bf16 NT GEMM for the fc layer of NetworkImpl (reference src/bin/train.cc:243-246,
// `sequential->push_back(torch::nn::Linear(3136, hidden_size))`) at training batch sizes.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        A: [M][K] bf16 (activations / dh), B: [N][K] bf16 (weights)
//
// The small-tile kernels in gemm.hpp stall once per 64-deep k-step on the `vmcnt(0)` their barrier emits
// while the next tile's LDS-DMA is still in flight: with 2 workgroups per CU they sit at ~11 % of the MFMA
// roof (PMC: 80 % of wave cycles parked in s_waitcnt).  This kernel keeps everything in flight ACROSS
// barriers instead:
//   * 128 x 128 x 64 tiles; 768 threads = 12 waves with SPECIALISED roles (round 3): 8 consumer waves (2 along m x 4
//     along n, 64 x 32 outputs per wave: fragment reads, MFMAs, the epilogue) and 4 loader waves that issue nothing but
//     the LDS-DMA of the ring, 8 x 1 KB per loader wave and stage.  The roles run SEPARATE loops with equal barrier
//     counts (one loop with role branches makes hipcc merge the two wait-count states and allocate the union of both
//     register sets: DESIGN.md 4a, "a compiler trap").  Rounds 1-2 had every wave do both jobs - 4 DMA pieces and their
//     address arithmetic between its 16 MFMAs: a stage took ~0.75 us where its DMA alone and its MFMAs alone each took
//     0.4-0.6 us.  Specialised: fc fwd 32.7 -> 28.4 us, fc dgrad 45.4 -> 38.6 us alone at 4096 samples, update 465 ->
//     454 us per minibatch (same box; 2 loader waves 30.7 / 40.8 us, 8 loader waves 67.6 / 96 us, wave priorities for
//     either role: no change, ring depth 3 / 4 / 5: 27.4 / 28.4 / 29.1 us);
//   * a ring of NST LDS stage buffers (32 KB each) filled by global_load_lds_dwordx4.  "Stage g + 1 has landed" is the
//     loaders' COUNTED `s_waitcnt vmcnt(8 * younger batches)` - never vmcnt(0) inside the loop - in front of the stage's
//     one raw s_barrier; "stage g's buffer is free" is every consumer's lgkmcnt(0) in front of the same barrier;
//   * the MFMA fragments are double-buffered in registers: while the 16 MFMAs of stage g run, the 12
//     ds_read_b128 of stage g + 1 are issued between them, so the LDS latency does not sit between a barrier and the
//     first MFMA;
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
//        tile is DMA'd into LDS by the consumer waves (an ordinary global load would be a second memory round trip per
//        lane), gated in place and written back with whole 256-byte rows.
//
// (A transposed-gather variant for the weight gradient - both operands k-major, ds_read_b64_tr_b16 fragments - was built
// and measured in rounds 1-2: 51-57 us against 45 us for gemm_tn_kernel; deleted in round 3.)
//
// Wait-count rule used throughout: `s_waitcnt vmcnt(N)` is safe iff N <= the number of vector-memory
// instructions this wave ISSUED after the batch it needs (they retire in issue order).
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
constexpr int PIPE_STAGE_CHUNKS = 2048; // 16-byte chunks per stage: A rows 0..127 x 8, then B rows 0..127 x 8
template <int MODE, int NST> constexpr size_t gemm_pipe_smem() {
  return (size_t)(NST * PIPE_STAGE_CHUNKS + (MODE == 1 ? 128 * 16 : 0)) * 16;
}

struct PipeFrags {
  u32x4 a[2][4], b[2][2]; // [k-chunk of 32][16-row atom]
};

constexpr int PIPE_THREADS = 768; // 8 consumer waves + 4 loader waves
template <int MODE, int NST> __global__ __launch_bounds__(PIPE_THREADS) void gemm_pipe_kernel(PipeParams P) {
  constexpr int NL = 4, PPL = 32 / NL; // loader waves; wave-pieces per loader wave and stage
  static_assert(NST >= 3, "ring depth");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  u32x4 *ring = reinterpret_cast<u32x4 *>(smem);
  u32x4 *etile = ring + NST * PIPE_STAGE_CHUNKS; // MODE 1: [128 rows][16 chunks], slot s of row r = chunk s ^ (r & 15)
  typedef __attribute__((address_space(1))) const void *gptr;
  typedef __attribute__((address_space(3))) void *lptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 8; // waves 8-11 stage, waves 0-7 multiply
  const int lw = wave - 8;
  const int fr = lane & 15, fg = lane >> 4;
  const int wm = (wave & 7) >> 2, wn = wave & 3;
  const int eslot = tid & 15, er0 = (tid & 511) >> 4; // gate / output tile: slot eslot of rows er0 + 32 i (consumers)

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

  int cord = 0, ck = 0;
  Job cjob = decode(0);
  if (loader) {
    // ================================================================= loader waves (8 - 11): nothing but LDS-DMA
    // stage s is issued NST stages ahead of its consumption; wave lw moves the wave-pieces lw + 4 q (q = 0..7) of a stage:
    // piece wp = 64 consecutive 16-byte chunks, chunk c = (wp & 15) * 64 + lane of operand wp >> 4 (0: A rows, 1: B rows)
    // is slot c & 7 of row c >> 3
    int pord = 0, pk = 0, pissued = 0, pbuf = 0;
    Job pjob = cjob;
    const bf16 *ps[PPL];
    auto set_src = [&]() {
#pragma unroll
      for (int q = 0; q < PPL; ++q) {
        const int wp = lw + NL * q, c = (wp & 15) * 64 + lane, r = c >> 3, sw = ((c & 7) ^ (r & 7)) * 8;
        ps[q] = wp < 16 ? P.A + (long)min(pjob.m0 + r, P.M - 1) * P.lda + (long)pjob.ks0 * 64 + sw
                        : P.B + (long)min(pjob.n0 + r, P.N - 1) * P.ldb + (long)pjob.ks0 * 64 + sw;
      }
    };
    set_src();
    auto issue = [&]() { // one batch = 8 LDS-DMA instructions of this wave
#pragma unroll
      for (int q = 0; q < PPL; ++q)
        __builtin_amdgcn_global_load_lds((gptr)(ps[q] + (long)pk * 64),
                                         (lptr)(ring + pbuf * PIPE_STAGE_CHUNKS + (lw + NL * q) * 64), 16, 0, 0);
      pbuf = pbuf + 1 == NST ? 0 : pbuf + 1;
      ++pissued;
      if (++pk == pjob.nks && ++pord < nj) {
        pjob = decode(pord);
        pk = 0;
        set_src();
      }
    };
    auto wait_stage_batches = [&](int batches) { // at most `batches` 8-instruction batches of this wave outstanding
      switch (batches) {
      case 0: wait_vm<0>(); break;
      case 1: wait_vm<PPL>(); break;
      case 2: wait_vm<2 * PPL>(); break;
      default: wait_vm<(3 * PPL < 63 ? 3 * PPL : 63)>(); break;
      }
    };
#pragma nounroll
    for (int i = 0; i < NST && pissued < G; ++i)
      issue();
    wait_stage_batches(min(NST - 1, G - 1));
    __builtin_amdgcn_s_barrier(); // stage 0 has landed
#pragma nounroll
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) // stage g + 1 has landed once only the batches issued after it are outstanding
        wait_stage_batches(min(NST - 2, G - 2 - g));
      __builtin_amdgcn_s_barrier(); // ... and every consumer has read stage g's fragments: its buffer is free
      if (pissued < G)
        issue();
      if (++ck == cjob.nks) { // the consumers' epilogue: meet them at its barriers
        if constexpr (MODE == 1) {
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_s_barrier();
        }
        ck = 0;
        if (++cord < nj)
          cjob = decode(cord);
      }
    }
    return;
  }

  // =================================================================== consumer waves (0 - 7): fragment reads + MFMAs
  int cbuf = 0; // ring buffer of the stage whose fragments are being READ (stage g + 1)
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
  PipeFrags F0, F1;
  { // stage 0 has landed (the barrier the loader waves reach after their counted wait): read its fragments
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

  // one stage: MFMAs on `cur` (stage g) while `nxt` (stage g + 1) is read
  auto stage = [&](int g, PipeFrags &cur, PipeFrags &nxt) {
    const bool has_next = g + 1 < G;
    // this wave's reads of stage g (issued one stage ago) are complete -> after the barrier its buffer may be refilled
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { // 4 x (4 MFMAs, 3 fragment reads)
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
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_next)
      cbuf = cbuf + 1 == NST ? 0 : cbuf + 1;
    if constexpr (MODE == 1) {
      if (ck == 0) { // gate tile of this job (its buffer was last read before the barrier above)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = er0 + 32 * i;
          const bf16 *src =
              P.gate + (long)min(cjob.m0 + r, P.M - 1) * P.N + min(cjob.n0 + ((eslot ^ (r & 15)) * 8), P.N - 8);
          __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(etile + wave * 64 + 512 * i), 16, 0, 0);
        }
      }
    }
    if (++ck == cjob.nks) { // ---------------- epilogue of this job
      const int mrow = cjob.m0 + wm * 64 + fr, ncol = cjob.n0 + wn * 32 + fg * 4;
      if constexpr (MODE == 0) {
        {
          float *out = P.out_f32 + (long)cjob.slice * P.M * P.N;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int n = ncol + j * 16;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (cjob.slice == 0 && n < P.N)
              b = *reinterpret_cast<const f32x4 *>(P.bias + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int m = mrow + i * 16;
              if (m < P.M && n < P.N)
                *reinterpret_cast<f32x4 *>(out + (long)m * P.N + n) = acc[i][j] + b;
            }
          }
        }
      } else {
        // (consumers issue no stage DMA here: everything of theirs that is outstanding is the gate tile and older stores)
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        {
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
        }
        __builtin_amdgcn_s_barrier();
        {
#pragma unroll
          for (int i = 0; i < 4; ++i) { // whole 256-byte rows: 16 lanes x 16 B
            const int r = er0 + 32 * i, m = cjob.m0 + r, n = cjob.n0 + ((eslot ^ (r & 15)) * 8);
            const u32x4 v = etile[r * 16 + eslot];
            if (m < P.M && n < P.N)
              *reinterpret_cast<u32x4 *>(P.out_bf16 + (long)m * P.N + n) = v;
          }
        }
      }
      zero_acc();
      ck = 0;
      if (++cord < nj)
        cjob = decode(cord);
    }
  };

#pragma nounroll
  for (int g = 0; g < G; g += 2) { // (without nounroll hipcc expands this loop into ~90k lines of ISA)
    stage(g, F0, F1);
    if (g + 1 < G)
      stage(g + 1, F1, F0);
  }
}

} // namespace aleppo
