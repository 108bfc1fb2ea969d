// kernels.hip - HBM-bound / latency-bound kernels of the PPO-over-ALE hot path for gfx950:
// frame ingest (LUT + area resize + 2-frame max + 4-frame stack + rollout-slot write), action head +
// categorical sampling, reward clamp + GAE + returns + old log-probs, PPO loss forward/backward fused
// with the head linear layers, split-K slab reduction, global-norm clip + Adam, and the layout
// conversions used only at the C-ABI boundary.  Each kernel cites the reference code it replaces.
#include "common.hpp"
#include "gemm.hpp" // bf16 / vector typedefs
#include <cstdlib>

namespace aleppo {

// Rollout-plane storage type RT: float (the reference's Buffer, buffer.cc:12-38) or IEEE half (BASELINE configs[4],
// "fp16 rollout buffer").  Arithmetic is always fp32: planes are widened on load and rounded (RNE) on store.
typedef _Float16 f16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_xor(v, o, 64);
  return v;
}
// deterministic block reduction (256 threads): wave shuffles, then wave 0 sums the 4 partials in order
__device__ __forceinline__ float block_sum_256(float v, float *s4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0)
    s4[wave] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// ================================================================================================
// Frame ingest.  Replaces, per env-step: gray LUT (environment.cc:48-55) + resize (vision.cc:86-95 on
// the env threads; here the area spec of vision.cc:8-32) + 2-frame max (max_and_skip.cc:33-42) +
// Rollout::update_observations (rollout.cc:184-196) + Buffer::add's observation copy (buffer.cc:47).
// The 4-frame stack of one pixel is ONE u32 (byte 0 = newest frame, Q11), so "shift + insert" is
// (old << 8) | new and "broadcast on episode start" is new * 0x01010101.  The stack for slot t+1
// is written straight into the rollout buffer; slot t is never copied again.
// grid (28 x E) workgroups of 256 threads, one thread per output pixel (a 4-pixels-per-thread variant with
// dword loads measured slower: 20 vs 17 us - fewer threads to hide the LUT read latency).
// ================================================================================================
template <bool RAW>
__global__ __launch_bounds__(256) void ingest_kernel(const uint8_t *__restrict__ frames, const uint8_t *__restrict__ lut,
                                                      const uint8_t *__restrict__ start, StartBits sbits, uint32_t *obs,
                                                      int slots, int t_src, int t_dst) {
  // one thread per output pixel: every load of a thread (6 raw dwords, the old packed pixel, the start flag)
  // is independent and issued up front - one memory round trip, no staging barrier on the frame data; the
  // 256-entry LUT sits in LDS.  A wave's 64 adjacent output pixels read ~122 adjacent raw bytes per row.
  const int e = blockIdx.y, tid = threadIdx.x;
  const int pix = blockIdx.x * 256 + tid;
  __shared__ uint8_t slut[256];
  const bool live = pix < FRAME_PIX;
  const int i = live ? pix / 84 : 0, j = live ? pix - i * 84 : 0;
  uint32_t old = 0;
  if (live)
    old = obs[((size_t)e * slots + t_src) * FRAME_PIX + pix];
  // episode-start flag: from the kernel-argument bitmask (no upload on the critical path) or from memory
  const bool st = start ? start[e] != 0 : ((sbits.w[e >> 5] >> (e & 31)) & 1u) != 0;
  uint32_t v = 0;
  if (RAW) {
    const int y0 = (i * RAW_H) / 84, x0 = (j * RAW_W) / 84, x1 = ((j + 1) * RAW_W + 83) / 84; // 3 rows, 2-3 cols
    const bool wide = (x1 - x0) == 3;
    // ONE unaligned dword per (frame, row) instead of 2-3 byte loads: the kernel is bound by the number of
    // vector-memory instructions (20 -> 8 per pixel).  The dword starts at min(x0, RAW_W - 4) so it never leaves
    // the row; the wanted bytes are shifted down.
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    const int xa = min(x0, RAW_W - 4), sh = (x0 - xa) * 8;
    uint32_t roww[2][3];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const uint8_t *src = frames + ((size_t)e * 2 + f) * (RAW_H * RAW_W) + (size_t)y0 * RAW_W + xa;
#pragma unroll
      for (int y = 0; y < 3; ++y)
        roww[f][y] = live ? *reinterpret_cast<const u32_unaligned *>(src + y * RAW_W) : 0u;
    }
    slut[tid] = lut[tid];
    __syncthreads();
    int best = 0;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      int s = 0;
#pragma unroll
      for (int y = 0; y < 3; ++y) {
        const uint32_t w = roww[f][y] >> sh;
        s += slut[w & 255u] + slut[(w >> 8) & 255u] + (wide ? slut[(w >> 16) & 255u] : 0);
      }
      // adaptive-average (area) mean in f32 like interpolate(mode=area), round-half-even to u8
      const int q = (int)rintf((float)s / (wide ? 9.0f : 6.0f));
      best = max(best, q);
    }
    v = (uint32_t)min(best, 255);
  } else if (live) {
    v = frames[(size_t)e * FRAME_PIX + pix];
  }
  if (live)
    obs[((size_t)e * slots + t_dst) * FRAME_PIX + pix] = st ? v * 0x01010101u : ((old << 8) | v);
}

void launch_ingest(hipStream_t s, bool raw, const uint8_t *frames, const uint8_t *lut, const uint8_t *start,
                   const StartBits *sbits, uint32_t *obs, int E, int slots, int t_src, int t_dst) {
  const dim3 g((FRAME_PIX + 255) / 256, E);
  StartBits sb{};
  if (sbits)
    sb = *sbits;
  if (raw)
    hipLaunchKernelGGL(ingest_kernel<true>, g, dim3(256), 0, s, frames, lut, start, sb, obs, slots, t_src, t_dst);
  else
    hipLaunchKernelGGL(ingest_kernel<false>, g, dim3(256), 0, s, frames, lut, start, sb, obs, slots, t_src, t_dst);
}

__global__ void copy_slot_kernel(uint32_t *obs, int slots, int src, int dst) {
  const int e = blockIdx.y;
  const u32x4 *s = reinterpret_cast<const u32x4 *>(obs + ((size_t)e * slots + src) * FRAME_PIX);
  u32x4 *d = reinterpret_cast<u32x4 *>(obs + ((size_t)e * slots + dst) * FRAME_PIX);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < FRAME_PIX / 4)
    d[i] = s[i];
}
void launch_copy_slot(hipStream_t s, uint32_t *obs, int E, int slots, int src, int dst) {
  hipLaunchKernelGGL(copy_slot_kernel, dim3((FRAME_PIX / 4 + 255) / 256, E), dim3(256), 0, s, obs, slots, src, dst);
}

// ================================================================================================
// The slot-ahead hand-off's gate (rollout.cc:204-208 / :312-313: the next forward pass may only start once the host has
// the actions and the emulators have delivered the frames).  The next slot's kernels are enqueued BEHIND this one-wave
// kernel; it returns when the host has stored a sequence number >= seq into the release word (mapped page-locked host
// memory, read with system-scope loads: every poll is a bus round trip, nothing is cached).
// Exit condition every wave reaches: after timeout_ticks of the 100 MHz wall clock the kernel gives up, reports the
// sequence number it was waiting for in go[1] and returns - the stream drains, the host finds the report and fails the
// context; the GPU never waits for ever on a host that has gone away.
// (Replaces hipStreamWaitValue32, which this runtime implements as the same kind of polling kernel - but without an
// exit condition, behind a capability flag, and on a 32-bit word that wraps.)
// ================================================================================================
__global__ __launch_bounds__(64) void gate_kernel(unsigned long long *go, unsigned long long seq,
                                                   unsigned long long timeout_ticks) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > timeout_ticks) {
        __hip_atomic_store(go + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
}
void launch_gate(hipStream_t s, unsigned long long *go_dev, unsigned long long seq, unsigned long long timeout_ticks) {
  hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, s, go_dev, seq, timeout_ticks);
}

// ================================================================================================
// Action head + categorical sampling (action selector closure, train.cc:367-379): logits / value from
// the hidden vector, softmax, multinomial(1, replacement) == argmax_k(p_k / q_k), q ~ Exp(1) (Q10),
// first maximum wins.  One wave per environment.  Actions go to the rollout slot AND to pinned host
// memory (replaces the per-env .item<int64_t>() of rollout.cc:312-313).
// ================================================================================================
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// h = sum of the fc split-K slices + fc bias, formed on the fly (all NSPLIT*8 loads of a lane are independent).
// Lane k < A owns action k (softmax term, noise, p/q); a wave arg-max picks the first maximum.
// Hand-off to the host: actions go to pinned memory, then ONE ticket word is published after every wave's
// stores are system-visible (fence + device counter, last wave publishes) - the host polls the ticket instead
// of synchronising the stream, so PCIe write latency overlaps the host's next enqueue.
// (Measured and rejected: one self-describing 8-byte word { ticket | action } per environment, no drain / counter /
// ticket store, the host sweeping all E words - 5.4-5.6 vs 4.9-5.3 ms per 128-slot rollout on the same box: the host
// then reads the very lines the device is still writing.)
// probs_in != nullptr (the stateless aleppo_sample operator, train.cc:374-375 alone): the head is skipped and lane k
// takes p_k from probs_in[e][k]; the division, the arg-max and the stores are the very same instructions.
template <int NSPLIT, class RT>
__global__ __launch_bounds__(256) void infer_head_kernel(const float *__restrict__ hpart, const float *__restrict__ bfc,
                                                          const float *__restrict__ Wh, const float *__restrict__ bh,
                                                          const float *__restrict__ noise, uint64_t seed,
                                                          uint64_t counter, RT *logits_t, RT *values_t,
                                                          int *actions_t, int64_t *pinned, unsigned int *done_ctr,
                                                          long long ticket, int E, int H, int A,
                                                          const float *__restrict__ probs_in) {
  extern __shared__ float sWh[]; // [(A+1)][H] head weights: ONE parallel round trip for the whole workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 4 + wave;
  // Lane l owns hidden units 4l .. 4l+3 and H/2 + 4l .. (H <= 512, H % 8 == 0): two 16-byte loads per split-K slice
  // instead of eight scalar ones (the kernel is latency-bound on ~70 vector-memory instructions per wave; now 18),
  // consecutive lanes on consecutive 16-byte pieces (coalesced, conflict-free LDS reads of the head weights).
  float hv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool heads = probs_in == nullptr; // workgroup-uniform
  if (heads) {
    f32x4 part[NSPLIT][2];
    const bool own = e < E && lane * 8 < H;
#pragma unroll
    for (int z = 0; z < NSPLIT; ++z) { // issue every split-K partial load first (independent)
      const float *src = hpart + ((size_t)z * E + (own ? e : 0)) * H + (own ? lane * 4 : 0);
      part[z][0] = *reinterpret_cast<const f32x4 *>(src);
      part[z][1] = *reinterpret_cast<const f32x4 *>(src + H / 2);
    }
    {
      const float *b = bfc + (own ? lane * 4 : 0);
      const f32x4 b0 = *reinterpret_cast<const f32x4 *>(b), b1 = *reinterpret_cast<const f32x4 *>(b + H / 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hv[i] = b0[i];
        hv[4 + i] = b1[i];
      }
    }
    for (int k = threadIdx.x; k < (A + 1) * H / 4; k += 256)
      reinterpret_cast<f32x4 *>(sWh)[k] = reinterpret_cast<const f32x4 *>(Wh)[k];
#pragma unroll
    for (int z = 0; z < NSPLIT; ++z)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hv[i] += part[z][0][i];
        hv[4 + i] += part[z][1][i];
      }
  }
  __syncthreads();
  if (e < E) {
    float zmine = 0.f; // lane a keeps logit a (a < A) / the value (a == A)
    for (int a = 0; heads && a <= A; ++a) {
      float s = 0.f;
      if (lane * 8 < H) {
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sWh + a * H + lane * 4);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sWh + a * H + H / 2 + lane * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          s += hv[i] * w0[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          s += hv[4 + i] * w1[i];
      }
      s = wave_sum(s) + bh[a];
      if (lane == a)
        zmine = s;
    }
    const bool isact = lane < A;
    float mx = isact ? zmine : -3.0e38f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
      mx = fmaxf(mx, __shfl_xor(mx, o, 64)); // A <= 18 < 32
    const float ex = isact ? expf(zmine - mx) : 0.f;
    float sum = ex;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
      sum += __shfl_xor(sum, o, 64);
    float q = 1.f;
    if (isact) {
      if (noise) {
        q = noise[(size_t)e * A + lane];
      } else { // Philox4x32-10 block (counter, env, lane/4): four actions share a block
        uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)e, (uint32_t)(lane >> 2)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t w = (lane & 3) == 0 ? c[0] : (lane & 3) == 1 ? c[1] : (lane & 3) == 2 ? c[2] : c[3];
        q = -logf(((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f)); // u in (0,1)
      }
    }
    float pk = ex / sum; // softmax (train.cc:374)
    if (!heads)
      pk = isact ? probs_in[(size_t)e * A + lane] : 0.f;
    float r = isact ? pk / q : -1.f; // p_k / q_k (train.cc:374-375)
    int best = lane;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { // arg-max, ties -> lowest index (first maximum wins)
      const float ro = __shfl_xor(r, o, 64);
      const int bo = __shfl_xor(best, o, 64);
      if (ro > r || (ro == r && bo < best)) {
        r = ro;
        best = bo;
      }
    }
    if (heads && isact)
      logits_t[(size_t)e * A + lane] = (RT)zmine;
    if (heads && lane == A)
      values_t[e] = (RT)zmine;
    if (lane == 0) {
      actions_t[e] = best;
      if (pinned) // system-scope (sc0 sc1, write-through) store straight to the mapped host buffer
        __hip_atomic_store(pinned + e, (int64_t)best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (done_ctr && pinned) {
    // Publish.  The host only needs the actions; they were written with system-scope stores, so a full
    // system-scope release (an L2 write-back of everything this rollout slot dirtied, ~4 us) is not needed:
    // every storing wave drains its stores (vmcnt(0): acknowledged), the workgroup meets, one thread bumps the
    // device counter and the last arriver - which therefore runs after every wave's acknowledged stores - writes
    // the ticket, again at system scope.  All atomics are RELAXED on purpose (an agent / system release would emit
    // the buffer_wbl2 this path exists to avoid); the order comes from the explicit vmcnt(0) drain, the barrier and
    // the data dependency of the ticket store on the counter's return value.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int prev = __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == gridDim.x - 1) {
        __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<long long *>(pinned + E), ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}
void launch_infer_head(hipStream_t s, const float *hpart, int nsplit, const float *bfc, const float *Wh,
                       const float *bh, const float *noise, uint64_t seed, uint64_t counter, void *logits_t,
                       void *values_t, int *actions_t, int64_t *pinned, unsigned int *done_ctr, long long ticket, int E,
                       int H, int A, const float *probs_in, bool rt16) {
  (void)nsplit; // always FC_SPLITS on the acting path
  const dim3 g((E + 3) / 4), b(256);
  const size_t sm = (size_t)(A + 1) * H * sizeof(float);
  if (rt16)
    hipLaunchKernelGGL((infer_head_kernel<FC_SPLITS, f16>), g, b, sm, s, hpart, bfc, Wh, bh, noise, seed, counter,
                       static_cast<f16 *>(logits_t), static_cast<f16 *>(values_t), actions_t, pinned, done_ctr, ticket, E,
                       H, A, probs_in);
  else
    hipLaunchKernelGGL((infer_head_kernel<FC_SPLITS, float>), g, b, sm, s, hpart, bfc, Wh, bh, noise, seed, counter,
                       static_cast<float *>(logits_t), static_cast<float *>(values_t), actions_t, pinned, done_ctr, ticket,
                       E, H, A, probs_in);
}

// ================================================================================================
// Reward clamp + GAE + returns + masks + old log-probs in ONE pass: Buffer::get (buffer.cc:58-77),
// ai::gae::gae (gae.cc:49-79) and prepare_batch's normalize_logits (train.cc:272-283).
// Rollout scalars are time-major [T][E] so each wave reads 64 consecutive environments per slot
// (coalesced); one thread owns one environment and scans t = T-1..0.  Outputs are the env-major
// (n = e*T + t) training arrays the minibatch slices index (Q1, Q5).  The float op order of
// gae.cc:61-66 is pinned (fp contract off) so no fma contraction changes the rounding.
// ================================================================================================
__device__ __forceinline__ float gae_step(float r, float v, float nv, float last, float gamma, float gl, bool st,
                                          bool te, bool tr) {
#pragma clang fp contract(off) // keep the reference's separate mul / add roundings (no fma)
  const float t1 = gamma * nv;
  const float t2 = r + t1;
  const float boot = t2 - v;                                           // (r + g*nv) - v
  const float t3 = gl * last;
  float a = boot + t3;                                                 // running,   gae.cc:61-63
  if (st)
    a = 0.f;                                                           // start,     gae.cc:68-69
  if (te)
    a = r - v;                                                         // terminal,  gae.cc:64,70-71
  if (tr)
    a = boot;                                                          // truncated, gae.cc:65-66,72-73
  return a;
}

// One thread = one environment, scanning t = T-1 .. 0.  The scan itself is a short dependent chain; what costs is
// memory latency, so the inputs of 16 time steps are loaded together and the NEXT 16 are already in flight (second
// register set) while a chunk is processed.  A chunk is processed branch-free (all 16 steps valid, the flag check
// accumulates into a register): with per-step branches the compiler sinks every load into its step's block and each
// step then pays a full memory round trip behind the previous step's scattered stores (54 us for T = 128).
struct GaeChunk {
  static constexpr int CH = 16;
  float r[CH], v[CH];
  uint8_t te[CH], tr[CH], st[CH];
};
template <class RT>
__global__ __launch_bounds__(64) void gae_kernel(uint8_t *rec, size_t rb, const RT *__restrict__ values_tm, RT *adv_n,
                                                  RT *ret_n, uint8_t *mask_n, int *err, int E, int T, float gamma,
                                                  float lambda, int clamp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E)
    return;
  constexpr int CH = GaeChunk::CH;
  const float gl = gamma * lambda;
  float last = 0.f, nv = (float)values_tm[(size_t)T * E + e];
  int bad = 0;
  auto step = [&](int t, float r, float v, bool bte, bool btr, bool bst) {
    // buffer.cc:67 clamp_, in place (Buffer::get).  clamp == 0: ai::gae::gae alone (the stateless aleppo_gae
    // operator) - rewards are used as given and left untouched.
    const float rc = clamp ? fminf(fmaxf(r, -1.0f), 1.0f) : r;
    if (clamp)
      reinterpret_cast<float *>(rec + (size_t)t * rb)[e] = rc;
    bad |= ((int)bte + (int)btr + (int)bst > 1) ? 1 : 0; // gae.cc:49-53
    const float a = gae_step(rc, v, nv, last, gamma, gl, bst, bte, btr);
    const size_t n = (size_t)e * T + t;
    adv_n[n] = (RT)a;        // (the recursion keeps the unrounded fp32 value)
    ret_n[n] = (RT)(a + v);  // buffer.cc:70-71
    mask_n[n] = bst ? 0 : 1; // buffer.cc:74
    last = a;
    nv = v;
  };
  auto load = [&](GaeChunk &c, int t1) { // steps t1-1 .. t1-CH, all valid
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int t = t1 - 1 - k;
      const uint8_t *fl = rec + (size_t)t * rb + 4 * (size_t)E;
      c.r[k] = reinterpret_cast<const float *>(rec + (size_t)t * rb)[e];
      c.v[k] = (float)values_tm[(size_t)t * E + e];
      c.te[k] = fl[e];
      c.tr[k] = fl[E + e];
      c.st[k] = fl[2 * E + e];
    }
  };
  auto process = [&](const GaeChunk &c, int t1) {
#pragma unroll
    for (int k = 0; k < CH; ++k)
      step(t1 - 1 - k, c.r[k], c.v[k], c.te[k] != 0, c.tr[k] != 0, c.st[k] != 0);
  };
  int t1 = T;
  for (int rem = T % CH; rem > 0; --rem) { // the ragged top of the horizon, one step at a time
    const int t = --t1;
    const uint8_t *fl = rec + (size_t)t * rb + 4 * (size_t)E;
    step(t, reinterpret_cast<const float *>(rec + (size_t)t * rb)[e], (float)values_tm[(size_t)t * E + e], fl[e] != 0,
         fl[E + e] != 0, fl[2 * E + e] != 0);
  }
  if (t1 > 0) { // t1 is a multiple of CH (uniform control flow: T is a kernel argument)
    GaeChunk ca, cb;
    load(ca, t1);
    while (true) {
      const bool more_b = t1 - CH > 0;
      if (more_b)
        load(cb, t1 - CH);
      process(ca, t1);
      if (!more_b)
        break;
      const bool more_a = t1 - 2 * CH > 0;
      if (more_a)
        load(ca, t1 - 2 * CH);
      process(cb, t1 - CH);
      if (!more_a)
        break;
      t1 -= 2 * CH;
    }
  }
  if (bad)
    *err = 1;
}
// the embarrassingly parallel part of prepare_batch (train.cc:272-283): old log-probs + actions, one thread
// per (t, e) slot, written env-major
template <class RT>
__global__ void oldlp_kernel(const RT *__restrict__ logits_tm, const int *__restrict__ actions_tm, RT *oldlp_n,
                             int *act_n, int E, int T, int A) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x; // time-major index t*E + e
  if (i >= (long)E * T)
    return;
  const int t = (int)(i / E), e = (int)(i - (long)t * E);
  const size_t n = (size_t)e * T + t;
  act_n[n] = actions_tm[i];
  const RT *z = logits_tm + i * A;
  float mx = (float)z[0];
  for (int k = 1; k < A; ++k)
    mx = fmaxf(mx, (float)z[k]);
  float s = 0.f;
  for (int k = 0; k < A; ++k)
    s += expf((float)z[k] - mx);
  const float lse = mx + logf(s);
  for (int k = 0; k < A; ++k)
    oldlp_n[n * A + k] = (RT)((float)z[k] - lse); // train.cc:279 normalize_logits
}
template <class RT>
static void launch_gae_t(hipStream_t s, uint8_t *step_rec, size_t rec_bytes, const void *values_tm, const void *logits_tm,
                         const int *actions_tm, void *adv_n, void *ret_n, void *oldlp_n, int *act_n, uint8_t *mask_n,
                         int *err, int E, int T, int A, float gamma, float lambda, bool clamp) {
  if (logits_tm) // (the stateless aleppo_gae operator has no logits / actions)
    hipLaunchKernelGGL(oldlp_kernel<RT>, dim3((unsigned)(((long)E * T + 255) / 256)), dim3(256), 0, s,
                       static_cast<const RT *>(logits_tm), actions_tm, static_cast<RT *>(oldlp_n), act_n, E, T, A);
  hipLaunchKernelGGL(gae_kernel<RT>, dim3((E + 63) / 64), dim3(64), 0, s, step_rec, rec_bytes,
                     static_cast<const RT *>(values_tm), static_cast<RT *>(adv_n), static_cast<RT *>(ret_n), mask_n, err,
                     E, T, gamma, lambda, clamp ? 1 : 0);
}
void launch_gae(hipStream_t s, uint8_t *step_rec, size_t rec_bytes, const void *values_tm, const void *logits_tm,
                const int *actions_tm, void *adv_n, void *ret_n, void *oldlp_n, int *act_n, uint8_t *mask_n, int *err,
                int E, int T, int A, float gamma, float lambda, bool clamp, bool rt16) {
  if (rt16)
    launch_gae_t<f16>(s, step_rec, rec_bytes, values_tm, logits_tm, actions_tm, adv_n, ret_n, oldlp_n, act_n, mask_n, err,
                      E, T, A, gamma, lambda, clamp);
  else
    launch_gae_t<float>(s, step_rec, rec_bytes, values_tm, logits_tm, actions_tm, adv_n, ret_n, oldlp_n, act_n, mask_n,
                        err, E, T, A, gamma, lambda, clamp);
}

// optional advantage normalisation over unmasked samples (NOT in the reference, Q2; off by default).
// phase 0: stats[0..2] += {sum, sumsq, count} (one block, deterministic); phase 1: apply.
template <class RT>
__global__ __launch_bounds__(256) void adv_norm_kernel(RT *adv, const uint8_t *mask, float *stats, long n, int phase) {
  __shared__ float s4[4];
  if (phase == 0) {
    float s = 0.f, q = 0.f, c = 0.f;
    for (long i = threadIdx.x; i < n; i += 256)
      if (mask[i]) {
        const float a = (float)adv[i];
        s += a;
        q += a * a;
        c += 1.f;
      }
    s = block_sum_256(s, s4);
    q = block_sum_256(q, s4);
    c = block_sum_256(c, s4);
    if (threadIdx.x == 0) {
      stats[0] = s;
      stats[1] = q;
      stats[2] = c;
    }
  } else {
    const float c = stats[2];
    if (!(c > 0.f)) // every sample masked: nothing to normalise (and nothing enters the loss)
      return;
    const float mean = stats[0] / c;
    const float var = fmaxf((stats[1] - c * mean * mean) / fmaxf(c - 1.f, 1.f), 0.f);
    const float inv = 1.0f / (sqrtf(var) + 1e-8f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
      adv[i] = (RT)(((float)adv[i] - mean) * inv);
  }
}
void launch_adv_norm(hipStream_t s, void *adv_n, const uint8_t *mask_n, float *stats, long n, int phase, bool rt16) {
  const dim3 g(phase == 0 ? 1 : (unsigned)((n + 255) / 256));
  if (rt16)
    hipLaunchKernelGGL(adv_norm_kernel<f16>, g, dim3(256), 0, s, static_cast<f16 *>(adv_n), mask_n, stats, n, phase);
  else
    hipLaunchKernelGGL(adv_norm_kernel<float>, g, dim3(256), 0, s, static_cast<float *>(adv_n), mask_n, stats, n, phase);
}

// boundary conversions of a rollout plane (aleppo_set_batch / aleppo_read_batch)
template <class S, class D> __global__ void cast_plane_kernel(const S *src, D *dst, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dst[i] = (D)(float)src[i];
}
void launch_plane_to_float(hipStream_t s, const void *src, float *dst, long n, bool rt16) {
  const dim3 g((unsigned)std::min<long>((n + 255) / 256, 1024));
  if (rt16)
    hipLaunchKernelGGL((cast_plane_kernel<f16, float>), g, dim3(256), 0, s, static_cast<const f16 *>(src), dst, n);
  else
    hipLaunchKernelGGL((cast_plane_kernel<float, float>), g, dim3(256), 0, s, static_cast<const float *>(src), dst, n);
}
void launch_plane_from_float(hipStream_t s, const float *src, void *dst, long n, bool rt16) {
  const dim3 g((unsigned)std::min<long>((n + 255) / 256, 1024));
  if (rt16)
    hipLaunchKernelGGL((cast_plane_kernel<float, f16>), g, dim3(256), 0, s, src, static_cast<f16 *>(dst), n);
  else
    hipLaunchKernelGGL((cast_plane_kernel<float, float>), g, dim3(256), 0, s, src, static_cast<float *>(dst), n);
}

// unmasked-sample count per minibatch (losses.cc:19 masks.sum()); block per minibatch
__global__ __launch_bounds__(256) void mask_count_kernel(const uint8_t *mask, float *counts, long B) {
  __shared__ float s4[4];
  const uint8_t *m = mask + (size_t)blockIdx.x * B;
  float c = 0.f;
  for (long i = threadIdx.x; i < B; i += 256)
    c += m[i] ? 1.f : 0.f;
  c = block_sum_256(c, s4);
  if (threadIdx.x == 0)
    counts[blockIdx.x] = c;
}
void launch_mask_count(hipStream_t s, const uint8_t *mask_n, float *counts, long B, int M) {
  hipLaunchKernelGGL(mask_count_kernel, dim3(M), dim3(256), 0, s, mask_n, counts, B);
}

// ================================================================================================
// PPO head, training.  Fuses: action/value linear layers (train.cc:245-253,262-263), normalize_logits
// (losses.cc:45-47), losses::compute forward (losses.cc:4-26) with its closed-form backward (SURVEY
// app. B: autograd's gradient of the masked-mean loss), the head dgrad (dh) and the head wgrad
// partials.  One wave per sample row, 4 waves per workgroup, rows strided over the grid; every
// workgroup writes ONE partial slab of head-weight gradients (summed later in fixed order).
// 1/N_m uses the GLOBAL unmasked count so that an all-reduce SUM over ranks yields the mean (8e).
// ================================================================================================
// Wide action sets (AMAX = 18: 19 x 8 wgrad accumulators per lane) run 4 waves per workgroup: one wave per SIMD may
// use the whole 512-entry register file; with 8 waves the 256-register cap spilled the accumulators (150 us vs 22).
template <class T, int AMAX, class RT>
__global__ __launch_bounds__(AMAX > 10 ? 256 : 512) void head_train_kernel(
    const float *__restrict__ h, const float *__restrict__ Wh, const float *__restrict__ bh,
    const int *__restrict__ act, const RT *__restrict__ oldlp, const RT *__restrict__ adv,
    const RT *__restrict__ ret, const uint8_t *__restrict__ mask, const float *__restrict__ mask_count, Hyper hp,
    T *dh, float *ps_total, float *ps_clipped, float *ps_value, float *ps_entropy, float *ps_ratio, float *slab_w,
    float *slab_b, long B, int H, int A, float *logits_out, float *values_out, int hparts) {
  constexpr int A1 = AMAX + 1, HPL = 8; // H <= 512: 8 hidden units per lane
  constexpr int NWV = AMAX > 10 ? 4 : 8; // waves per workgroup
  extern __shared__ float smem[];
  float *sW = smem;                    // [(A+1)][H]
  float *sAcc = smem + (size_t)A1 * H; // [(A+1)][H] cross-wave wgrad accumulator
  float *sB = sAcc + (size_t)(A1 > NWV ? A1 : NWV) * H; // [NWV][A1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long rows_per_blk = (B + gridDim.x - 1) / gridDim.x;
  const long row0 = (long)blockIdx.x * rows_per_blk, row1 = min(B, row0 + rows_per_blk);
  // Everything a row needs (h, action, advantage, return, mask and ALL A old log-probs, so that nothing is a
  // dependent load) is fetched ONE ROW AHEAD: a row's memory round trips hide behind the previous row's math.
  // The first row's loads are issued before the weight staging below.
  float hnext[HPL], olp_n[AMAX], adv_n = 0.f, ret_n = 0.f;
  int act_n = 0;
  bool mask_n = false;
  auto fetch = [&](long r) {
    const bool ok = r < row1;
    // lane l owns hidden units 4l .. 4l+3 and H/2 + 4l .. H/2 + 4l+3 (H % 8 == 0): 16-byte loads instead of scalar
    // ones, and consecutive lanes touch consecutive 16-byte pieces (coalesced; conflict-free LDS reads of the weights)
#pragma unroll
    for (int i = 0; i < HPL; ++i)
      hnext[i] = 0.f;
    if (ok && lane * 8 < H) { // h arrives as `hparts` split-K partial slabs [hparts][B][H] (slab 0 carries the bias)
      for (int p = 0; p < hparts; ++p) {
        const float *src = h + ((size_t)p * B + r) * H + lane * 4;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + H / 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          hnext[i] += v0[i];
          hnext[4 + i] += v1[i];
        }
      }
    }
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      olp_n[a] = (ok && a < A) ? (float)oldlp[(size_t)r * A + a] : 0.f;
    act_n = ok ? act[r] : 0;
    adv_n = ok ? (float)adv[r] : 0.f;
    ret_n = ok ? (float)ret[r] : 0.f;
    mask_n = ok ? mask[r] != 0 : false;
  };
  fetch(row0 + wave);
  for (int i = tid; i < (A + 1) * H; i += 64 * NWV)
    sW[i] = Wh[i];
  __syncthreads();
  const float inv_nm = 1.0f / mask_count[0];
  float gW[A1][HPL], gb[A1];
#pragma unroll
  for (int a = 0; a < A1; ++a) {
    gb[a] = 0.f;
#pragma unroll
    for (int i = 0; i < HPL; ++i)
      gW[a][i] = 0.f;
  }
  for (long row = row0 + wave; row < row1; row += NWV) {
    float hv[HPL], olp_c[AMAX];
#pragma unroll
    for (int i = 0; i < HPL; ++i)
      hv[i] = hnext[i];
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      olp_c[a] = olp_n[a];
    const int ai = act_n;
    const float advi = adv_n, reti = ret_n;
    const bool maski = mask_n;
    fetch(row + NWV); // next row of this wave
    float z[A1];
#pragma unroll
    for (int a = 0; a < A1; ++a) {
      float s = 0.f;
      if (a <= A) {
        if (lane * 8 < H) {
          const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sW + a * H + lane * 4);
          const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sW + a * H + H / 2 + lane * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            s += hv[i] * w0[i];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            s += hv[4 + i] * w1[i];
        }
        s = wave_sum(s) + bh[a];
      }
      z[a] = s;
    }
    // every lane now holds logits z[0..A-1] and the value z[A]
    const float value = [&] {
      float v = 0.f;
#pragma unroll
      for (int a = 0; a < A1; ++a)
        if (a == A)
          v = z[a];
      return v;
    }();
    float mx = -3.0e38f;
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      if (a < A)
        mx = fmaxf(mx, z[a]);
    float se = 0.f;
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      if (a < A)
        se += expf(z[a] - mx);
    const float lse = mx + logf(se);
    float lp[AMAX], p[AMAX], ent = 0.f, lpa = 0.f, olpa = 0.f;
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
      lp[a] = 0.f;
      p[a] = 0.f;
      if (a < A) {
        lp[a] = z[a] - lse;            // losses.cc:45-47
        p[a] = expf(lp[a]);
        ent += p[a] * lp[a];           // losses.cc:41-43
        if (a == ai) {
          lpa = lp[a];
          olpa = olp_c[a];
        }
      }
    }
    ent = -ent;
    const float rho = expf(lpa - olpa);                                  // losses.cc:33
    const float crho = fminf(fmaxf(rho, 1.0f - hp.clip), 1.0f + hp.clip); // losses.cc:34-35
    const float un = rho * advi, cl = crho * advi;
    const float obj = fminf(un, cl);                                     // losses.cc:38
    const float dv = value - reti;
    const float lv = 0.5f * (dv * dv);                                   // losses.cc:15
    const float Ltot = -obj + hp.c_v * lv - hp.c_e * ent;                // losses.cc:17-18
    const float m = maski ? inv_nm : 0.f;                                // losses.cc:19 masked mean
    const bool active = advi >= 0.f ? (rho <= 1.0f + hp.clip) : (rho >= 1.0f - hp.clip);
    const float gs = active ? -rho * advi : 0.f;
    float dz[A1];
#pragma unroll
    for (int a = 0; a < A1; ++a) {
      dz[a] = 0.f;
      if (a < A && a < AMAX)
        dz[a] = m * (gs * ((a == ai ? 1.0f : 0.0f) - p[a < AMAX ? a : 0]) +
                     hp.c_e * p[a < AMAX ? a : 0] * (lp[a < AMAX ? a : 0] + ent));
      if (a == A)
        dz[a] = m * hp.c_v * dv;
    }
    if (lane == 0) {
      ps_total[row] = Ltot;
      ps_clipped[row] = obj;
      ps_value[row] = lv;
      ps_entropy[row] = ent;
      ps_ratio[row] = rho;
      if (logits_out) {
#pragma unroll
        for (int a = 0; a < AMAX; ++a)
          if (a < A)
            logits_out[(size_t)row * A + a] = z[a];
        values_out[row] = value;
      }
    }
    // head dgrad: dh = sum_a dz[a] * W[a][:]   and wgrad partial: gW[a][:] += dz[a] * h
    if (lane * 8 < H) {
      float d[HPL];
#pragma unroll
      for (int i = 0; i < HPL; ++i)
        d[i] = 0.f;
#pragma unroll
      for (int a = 0; a < A1; ++a)
        if (a <= A) {
          const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sW + a * H + lane * 4);
          const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sW + a * H + H / 2 + lane * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            d[i] += dz[a] * w0[i];
            d[4 + i] += dz[a] * w1[i];
          }
#pragma unroll
          for (int i = 0; i < HPL; ++i)
            gW[a][i] += dz[a] * hv[i];
        }
      T dr[HPL];
#pragma unroll
      for (int i = 0; i < HPL; ++i)
        dr[i] = (T)d[i];
      T *dst = dh + (size_t)row * H + lane * 4;
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<u32x2 *>(dst) = reinterpret_cast<const u32x2 *>(dr)[0];
        *reinterpret_cast<u32x2 *>(dst + H / 2) = reinterpret_cast<const u32x2 *>(dr)[1];
      } else {
        *reinterpret_cast<u32x4 *>(dst) = reinterpret_cast<const u32x4 *>(dr)[0];
        *reinterpret_cast<u32x4 *>(dst + H / 2) = reinterpret_cast<const u32x4 *>(dr)[1];
      }
    }
#pragma unroll
    for (int a = 0; a < A1; ++a)
      gb[a] += dz[a];
  }
  // deterministic cross-wave reduction, one head row at a time: every wave writes its partial of row a, then
  // thread j adds the NWV partials of column j in fixed order and stores the workgroup's slab entry
  float *sPart = sAcc; // [NWV][H] (reuses the accumulator region: (A+1)*H >= ... is not needed, H*NWV floats)
  float *ow = slab_w + (size_t)blockIdx.x * (A + 1) * H;
#pragma unroll
  for (int a = 0; a < A1; ++a) {
    if (a <= A) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < HPL; ++i) {
        const int j = (i < 4 ? 0 : H / 2) + lane * 4 + (i & 3);
        if (lane * 8 < H)
          sPart[wave * H + j] = gW[a][i];
      }
      if (lane == 0)
        sB[wave * A1 + a] = gb[a];
      __syncthreads();
      for (int j = tid; j < H; j += 64 * NWV) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w)
          sum += sPart[w * H + j];
        ow[a * H + j] = sum;
      }
    }
  }
  __syncthreads();
  if (tid <= A) {
    float sb = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w)
      sb += sB[w * A1 + tid];
    slab_b[(size_t)blockIdx.x * (A + 1) + tid] = sb;
  }
}

template <class T, class RT>
static void head_train_t(hipStream_t s, const float *h, const float *Wh, const float *bh, const int *act,
                         const RT *oldlp, const RT *adv, const RT *ret, const uint8_t *mask,
                         const float *mask_count, Hyper hp, void *dh, float *ps_total, float *ps_clipped,
                         float *ps_value, float *ps_entropy, float *ps_ratio, float *slab_w, float *slab_b, int nblk,
                         long B, int H, int A, float *lo, float *vo, int hparts) {
#define LAUNCH_HEAD(AM)                                                                                                \
  do {                                                                                                                 \
    const size_t sm = ((size_t)((AM + 1) + ((AM + 1) > 8 ? (AM + 1) : 8)) * H + 8 * (AM + 1)) * sizeof(float);        \
    if (sm > 48 * 1024)                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&head_train_kernel<T, AM, RT>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);                                  \
    hipLaunchKernelGGL((head_train_kernel<T, AM, RT>), dim3(nblk), dim3(AM > 10 ? 256 : 512), sm, s, h, Wh, bh, act, oldlp, adv, ret,  \
                       mask, mask_count, hp, static_cast<T *>(dh), ps_total, ps_clipped, ps_value, ps_entropy,         \
                       ps_ratio, slab_w, slab_b, B, H, A, lo, vo, hparts);                                                   \
  } while (0)
  if (A <= 4)
    LAUNCH_HEAD(4);
  else if (A <= 6)
    LAUNCH_HEAD(6);
  else if (A <= 10)
    LAUNCH_HEAD(10);
  else
    LAUNCH_HEAD(18);
#undef LAUNCH_HEAD
}
void launch_head_train(hipStream_t s, const float *h, const float *Wh, const float *bh, const int *act,
                       const void *oldlp, const void *adv, const void *ret, const uint8_t *mask,
                       const float *mask_count, Hyper hp, void *dh, int prec, float *ps_total, float *ps_clipped,
                       float *ps_value, float *ps_entropy, float *ps_ratio, float *slab_w, float *slab_b, int nblk,
                       long B, int H, int A, float *logits_out, float *values_out, int hparts, bool rt16) {
#define HEAD_ARGS(RT)                                                                                                  \
  s, h, Wh, bh, act, static_cast<const RT *>(oldlp), static_cast<const RT *>(adv), static_cast<const RT *>(ret), mask, \
      mask_count, hp, dh, ps_total, ps_clipped, ps_value, ps_entropy, ps_ratio, slab_w, slab_b, nblk, B, H, A,         \
      logits_out, values_out, hparts
  if (prec == ALEPPO_BF16) {
    if (rt16)
      head_train_t<bf16, f16>(HEAD_ARGS(f16));
    else
      head_train_t<bf16, float>(HEAD_ARGS(float));
  } else {
    if (rt16)
      head_train_t<float, f16>(HEAD_ARGS(f16));
    else
      head_train_t<float, float>(HEAD_ARGS(float));
  }
#undef HEAD_ARGS
}

// ================================================================================================
// Split-K slab reduction -> flat gradient (fixed summation order => run-to-run deterministic).
// ================================================================================================
struct ReduceArgs {
  ReduceSeg seg[12];
  int cprefix[13]; // prefix of workgroups per segment
  int wide[12];    // segment runs the 16-byte path (1024 outputs per workgroup)
  int nseg;
};
// Two shapes of work in one launch, chosen per segment on the host:
//  * narrow (small n, many slabs - the conv wgrads): one workgroup per 64 consecutive outputs, thread
//    (o = tid&63, q = tid>>6) sums slabs q, q+4, ... and the four partials are combined in fixed order
//    -> 4x the memory-level parallelism of a serial scan;
//  * wide (the fc weight: 1.6 M outputs, <= 4 slabs): one workgroup per 1024 outputs, every thread sums a
//    float4 column with 16-byte loads, keeping the same four interleaved partial sums per output.
// Both give the value ((s0+s4+..) + (s1+s5+..)) + ((s2+..) + (s3+..)) -> identical bits, run-to-run deterministic.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(ReduceArgs a, float *G) {
  __shared__ float part[4][64];
  int s = 0;
  while ((int)blockIdx.x >= a.cprefix[s + 1])
    ++s;
  const long n = a.seg[s].n;
  const int S = a.seg[s].S;
  if (a.wide[s]) { // workgroup-uniform
    const long j = ((long)((int)blockIdx.x - a.cprefix[s]) * 256 + threadIdx.x) * 4;
    if (j >= n)
      return;
    const float *p = a.seg[s].slab + j;
    float4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= S; k += 4) {
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        v[q] = *reinterpret_cast<const float4 *>(p + (long)(k + q) * n);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q].x += v[q].x;
        acc[q].y += v[q].y;
        acc[q].z += v[q].z;
        acc[q].w += v[q].w;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (k + q < S) {
        const float4 v = *reinterpret_cast<const float4 *>(p + (long)(k + q) * n);
        acc[q].x += v.x;
        acc[q].y += v.y;
        acc[q].z += v.z;
        acc[q].w += v.w;
      }
    float4 r;
    r.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x);
    r.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
    r.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z);
    r.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
    *reinterpret_cast<float4 *>(G + a.seg[s].dst + j) = r;
    return;
  }
  const long j = (long)((int)blockIdx.x - a.cprefix[s]) * 64 + (threadIdx.x & 63);
  const int q = threadIdx.x >> 6;
  float acc = 0.f;
  if (j < n) {
    const float *p = a.seg[s].slab + j;
#pragma unroll 4
    for (int k = q; k < S; k += 4)
      acc += p[(long)k * n];
  }
  part[q][threadIdx.x & 63] = acc;
  __syncthreads();
  if (q == 0 && j < n)
    G[a.seg[s].dst + j] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
void launch_reduce_slabs(hipStream_t s, const ReduceSeg *segs, int nseg, float *G) {
  static const long wide_min = [] { // A/B switch: smallest segment that takes the 16-byte path
    const char *e = getenv("ALEPPO_REDUCE_WIDE_MIN");
    return e ? atol(e) : 1L << 18;
  }();
  ReduceArgs a;
  a.nseg = nseg;
  a.cprefix[0] = 0;
  for (int i = 0; i < nseg; ++i) {
    a.seg[i] = segs[i];
    const bool aligned = segs[i].n % 4 == 0 && segs[i].dst % 4 == 0 && ((uintptr_t)segs[i].slab & 15) == 0 &&
                         ((uintptr_t)G & 15) == 0;
    a.wide[i] = aligned && segs[i].n >= wide_min;
    a.cprefix[i + 1] = a.cprefix[i] + (int)(a.wide[i] ? (segs[i].n + 1023) / 1024 : (segs[i].n + 63) / 64);
  }
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(a.cprefix[nseg]), dim3(256), 0, s, a, G);
}

// ================================================================================================
// clip_grad_norm_ (train.cc:12-46) + torch::optim::Adam (eps 1e-5, train.cc:360-362) in two kernels:
// per-block sum of squares, then every Adam block re-reduces the <=1024 partials in the same order
// (identical norm everywhere), scales by min(1, max_norm/(norm+1e-6)) and applies the update.  The
// bf16 compute copy of the weights is refreshed in the same pass.
// ================================================================================================
// Blocks [0, nmain) square G[0, n).  The blocks after them own the LAST gradient tensors (conv1 weight + bias, 64
// outputs per block) and either sum their split-K slabs first - the update's last slab reduce fused into this pass, one
// launch less on the serial tail of every minibatch - or, when the slabs were reduced before (data parallelism: the
// all-reduce needs G complete), read G.  Same partition and same summation order in both modes: identical partials.
struct SumsqTail {
  ReduceSeg seg[2]; // slab == nullptr: the tensor is already in G
  int chunks[2];    // 64-output chunks per tensor
};
__global__ __launch_bounds__(256) void sumsq_kernel(float *__restrict__ G, long n, float *partials, int nmain,
                                                    SumsqTail tail, int boff) {
  __shared__ float s4[4];
  __shared__ float part[4][64];
  float s = 0.f;
  const int blk = (int)blockIdx.x + boff; // (a launch may cover only the main blocks or only the tail blocks)
  if (blk < nmain) {
    const long per = (n + nmain - 1) / nmain;
    const long b = (long)blk * per, e = min(n, b + per);
    for (long i = b + threadIdx.x; i < e; i += 256)
      s += G[i] * G[i];
  } else {
    int c = blk - nmain, t = 0;
    if (c >= tail.chunks[0]) {
      c -= tail.chunks[0];
      t = 1;
    }
    const ReduceSeg sg = tail.seg[t];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long j = (long)c * 64 + o;
    float g = 0.f;
    if (sg.slab) { // workgroup-uniform; same order as reduce_slabs_kernel's narrow path
      float acc = 0.f;
      if (j < sg.n) {
        const float *p = sg.slab + j;
#pragma unroll 4
        for (int k = q; k < sg.S; k += 4)
          acc += p[(long)k * sg.n];
      }
      part[q][o] = acc;
      __syncthreads();
      if (q == 0 && j < sg.n) {
        g = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
        G[sg.dst + j] = g;
      }
    } else if (q == 0 && j < sg.n) {
      g = G[sg.dst + j];
    }
    s = g * g;
  }
  s = block_sum_256(s, s4);
  if (threadIdx.x == 0)
    partials[blk] = s;
}
// which: 0 = every block; 1 = only the blocks of G[0, n_main); 2 = only the tail tensors' blocks (same partials, same
// order).  Splitting the pass - main blocks on the weight-gradient stream beside conv1 wgrad, tail blocks after the join -
// measured slower (454.7 vs 450.0 us per minibatch: the tail launch alone still costs 7.8 us and the join gap stays).
int launch_sumsq(hipStream_t s, float *G, long n_main, float *partials, int nblk_main, const ReduceSeg tail[2],
                 int which) {
  SumsqTail t;
  for (int i = 0; i < 2; ++i) {
    t.seg[i] = tail[i];
    t.chunks[i] = (int)((tail[i].n + 63) / 64);
  }
  const int ntail = t.chunks[0] + t.chunks[1], nblk = nblk_main + ntail;
  const int first = which == 2 ? nblk_main : 0, count = which == 1 ? nblk_main : nblk - first;
  hipLaunchKernelGGL(sumsq_kernel, dim3(count), dim3(256), 0, s, G, n_main, partials, nblk_main, t, first);
  return nblk; // partials written once every block has run
}

struct long4_ranges {
  long begin[4], len[4];
};
// One Adam step of element i (train.cc:42-44 clip scale always applied; torch::optim::Adam's update form)
struct AdamScalars {
  float coef, step_size, bc2_sqrt, beta1, beta2, omb1, omb2, eps;
};
__device__ __forceinline__ float adam_element(long i, float *P, const float *__restrict__ G, float *Gs, float *M1,
                                              float *M2, const AdamScalars &a) {
  const float g = G[i] * a.coef;
  const float m = M1[i] * a.beta1 + a.omb1 * g;
  const float v = M2[i] * a.beta2 + a.omb2 * (g * g);
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  const float p = P[i] - a.step_size * (m / denom);
  M1[i] = m;
  M2[i] = v;
  P[i] = p;
  if (Gs)
    Gs[i] = g;
  return p;
}
// The dgrad-side weight layouts are TRANSPOSES of three weight tensors (W3d[c][(tap,oc)], W2d[class][c][(ab,oc)],
// WfcT[j][o]): the blocks that own those tensors walk them in 64-row tiles, write the updated weight to the compute copy
// in place and - through an LDS transpose, so that both sides are whole lines - to its transposed home.  (Until round 3
// two extra kernels re-read P on a side stream after every optimizer step: 28 us and 9.6 MB per step.)
struct AdamTiles {
  // tile t of tensor k: rows r0..r0+63 (row stride rs), cols c0..c0+cols-1 contiguous at src = off + r*rs + c;
  // transposed element (c, r) at dst + c*ds + r
  long off[3];      // Wfc, W3, W2 offsets in the flat vector
  int first[4];     // first tile index of each tensor (+ total)
  int H;
};
template <class T>
__global__ __launch_bounds__(256) void adam_kernel(float *P, const float *__restrict__ G, float *Gs, float *M1, float *M2,
                                                    T *Pc, T *WfcT, T *W3d, T *W2d, AdamTiles tl, long n_flat,
                                                    long4_ranges fr, const float *__restrict__ partials, int nblk,
                                                    float max_norm, const float *__restrict__ sched, float beta1,
                                                    float beta2, float eps, float *grad_norm_out) {
  __shared__ float s4[4];
  __shared__ float tile[64][65];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256)
    s += partials[i];
  s = block_sum_256(s, s4);
  const float norm = sqrtf(s);
  float coef = max_norm / (norm + 1e-6f); // train.cc:39
  coef = fminf(coef, 1.0f);               // train.cc:40-41
  if (blockIdx.x == 0 && threadIdx.x == 0 && grad_norm_out)
    *grad_norm_out = norm;                // pre-clip norm is what the reference reports (Q9)
  // lr / (1 - beta1^t) and sqrt(1 - beta2^t) of THIS optimizer step: device scalars (a captured hipGraph of the update
  // follows the annealed rate and the step count; the reference's captured graph bakes both, train.h:163-195)
  const AdamScalars a{coef, sched[0], sched[1], beta1, beta2, 1.0f - beta1, 1.0f - beta2, eps};
  const int ntile = tl.first[3];
  if ((int)blockIdx.x < ntile) { // a 64-row tile of Wfc / W3 / W2
    const int t = blockIdx.x;
    long src;      // flat index of tile element (0, 0)
    int rs, rows, cols;
    T *dst;        // transposed element (c, r) at dst[c * ds + r]
    int ds;
    if (t < tl.first[1]) {          // Wfc[o][j] -> WfcT[j][o]: tile (o-block, j-block of 64; 3136 = 49 * 64)
      const int ob = t / 49, jb = t - ob * 49;
      rs = FC_IN;
      rows = min(64, tl.H - ob * 64);
      cols = 64;
      src = tl.off[0] + (long)ob * 64 * FC_IN + jb * 64;
      dst = WfcT + (long)jb * 64 * tl.H + ob * 64;
      ds = tl.H;
    } else if (t < tl.first[2]) {   // W3[oc][tap][c] -> W3d[c][tap][oc]: one tile per tap
      const int tap = t - tl.first[1];
      rs = 576;
      rows = 64;
      cols = 64;
      src = tl.off[1] + tap * 64;
      dst = W3d + tap * 64;
      ds = 576;
    } else {                        // W2[oc][(kh,kw)][c] -> W2d[class][c][(ab)][oc]: one 64 x 32 tile per (kh, kw)
      const int k = t - tl.first[2], kh = k >> 2, kw = k & 3;
      const int cls = (kh & 1) * 2 + (kw & 1), ab = (kh >> 1) * 2 + (kw >> 1);
      rs = 512;
      rows = 64;
      cols = 32;
      src = tl.off[2] + k * 32;
      dst = W2d + cls * (32 * 256) + ab * 64;
      ds = 256;
    }
    const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
    if (c < cols) {
      // all 64 loads of a thread's 16 elements are issued before the first store (P / M1 / M2 are read and written through
      // the same pointers, so the compiler may not hoist them itself; a dependent load-compute-store chain per element made
      // this kernel latency-bound: 23 vs 13 us)
      float g[16], m1[16], m2[16], p0[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int r = r4 + 4 * k;
        const long i = src + (long)(r < rows ? r : 0) * rs + c;
        g[k] = G[i];
        m1[k] = M1[i];
        m2[k] = M2[i];
        p0[k] = P[i];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int r = r4 + 4 * k;
        if (r < rows) {
          const long i = src + (long)r * rs + c;
          const float gg = g[k] * a.coef;
          const float m = m1[k] * a.beta1 + a.omb1 * gg;
          const float v = m2[k] * a.beta2 + a.omb2 * (gg * gg);
          const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
          const float p = p0[k] - a.step_size * (m / denom);
          M1[i] = m;
          M2[i] = v;
          P[i] = p;
          if (Gs)
            Gs[i] = gg;
          if (Pc)
            Pc[i] = (T)p;
          tile[r][c] = p;
        }
      }
    }
    __syncthreads();
    const int r = threadIdx.x & 63, c4 = threadIdx.x >> 6;
    if (r < rows)
      for (int cc = c4; cc < cols; cc += 4)
        dst[(long)cc * ds + r] = (T)tile[r][cc];
    return;
  }
  // everything else, flat: index k of the compacted space of the (at most four) ranges between the tiled tensors
  const long stride = (long)(gridDim.x - ntile) * 256;
  for (long k = (long)((int)blockIdx.x - ntile) * 256 + threadIdx.x; k < n_flat; k += stride) {
    long i = k;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (i < fr.len[q]) {
        i += fr.begin[q];
        break;
      }
      i -= fr.len[q];
    }
    const float p = adam_element(i, P, G, Gs, M1, M2, a);
    if (Pc)
      Pc[i] = (T)p;
  }
}
void launch_adam(hipStream_t s, float *P, const float *G_in, float *G_out_scaled, float *M1, float *M2, void *Pc,
                 void *WfcT, void *W3d, void *W2d, const ParamLayout &L, int prec, const float *partials, int nblk,
                 float max_norm, const float *sched, float beta1, float beta2, float eps, float *grad_norm_out) {
  AdamTiles tl;
  tl.off[0] = (long)L.off[P_WFC];
  tl.off[1] = (long)L.off[P_W3];
  tl.off[2] = (long)L.off[P_W2];
  tl.H = L.H;
  tl.first[0] = 0;
  tl.first[1] = (L.H + 63) / 64 * 49;
  tl.first[2] = tl.first[1] + 9;
  tl.first[3] = tl.first[2] + 16;
  // flat ranges: [0, Wfc) heads, [bfc, W3), [b3, W2), [b2, end) (pads between tensors included: zero and stay zero)
  long4_ranges fr;
  const long edges[8] = {0, (long)L.off[P_WFC], (long)L.off[P_BFC], (long)L.off[P_W3], (long)L.off[P_B3],
                         (long)L.off[P_W2], (long)L.off[P_B2], (long)L.total()};
  long n_flat = 0;
  for (int q = 0; q < 4; ++q) {
    fr.begin[q] = edges[2 * q];
    fr.len[q] = edges[2 * q + 1] - edges[2 * q];
    n_flat += fr.len[q];
  }
  const int nb = tl.first[3] + (int)std::min<long>((n_flat + 255) / 256, 1024);
  if (prec == ALEPPO_BF16)
    hipLaunchKernelGGL(adam_kernel<bf16>, dim3(nb), dim3(256), 0, s, P, G_in, G_out_scaled, M1, M2,
                       static_cast<bf16 *>(Pc), static_cast<bf16 *>(WfcT), static_cast<bf16 *>(W3d),
                       static_cast<bf16 *>(W2d), tl, n_flat, fr, partials, nblk, max_norm, sched, beta1, beta2, eps,
                       grad_norm_out);
  else
    hipLaunchKernelGGL(adam_kernel<float>, dim3(nb), dim3(256), 0, s, P, G_in, G_out_scaled, M1, M2,
                       static_cast<float *>(nullptr), static_cast<float *>(WfcT), static_cast<float *>(W3d),
                       static_cast<float *>(W2d), tl, n_flat, fr, partials, nblk, max_norm, sched, beta1, beta2, eps,
                       grad_norm_out);
}

__global__ void cast_params_kernel(const float *P, bf16 *Pc, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    Pc[i] = (bf16)P[i];
}
void launch_cast_params(hipStream_t s, const float *P, void *Pc, long n) {
  hipLaunchKernelGGL(cast_params_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, s, P,
                     static_cast<bf16 *>(Pc), n);
}

// dgrad-side weight copies: W3d[c][(kh,kw,oc)], W2d[class][c][(a,b,oc)], WfcT[j][o]   (all in T)
template <class T> __global__ void pack_conv_dgrad_kernel(const float *W3, const float *W2, T *W3d, T *W2d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 64 * 576) {
    const int c = i / 576, rem = i - c * 576, tap = rem >> 6, oc = rem & 63;
    W3d[i] = (T)W3[oc * 576 + tap * 64 + c];
  } else if (i < 64 * 576 + 4 * 32 * 256) {
    const int k = i - 64 * 576;
    const int cls = k / (32 * 256), c = (k / 256) % 32, rem = k & 255, ab = rem >> 6, oc = rem & 63;
    const int kh = (cls >> 1) + 2 * (ab >> 1), kw = (cls & 1) + 2 * (ab & 1);
    W2d[k] = (T)W2[oc * 512 + (kh * 4 + kw) * 32 + c];
  }
}
template <class T> __global__ void transpose_cast_kernel(const float *in, T *out, int R, int C) { // out[C][R]
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32x8
  for (int k = ty; k < 32; k += 8)
    if (r0 + k < R && c0 + tx < C)
      tile[k][tx] = in[(size_t)(r0 + k) * C + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < C && r0 + tx < R)
      out[(size_t)(c0 + k) * R + r0 + tx] = (T)tile[tx][k];
}
void launch_pack_dgrad(hipStream_t s, const float *P, const ParamLayout &L, void *W2d, void *W3d, void *WfcT,
                       int prec) {
  const int n = 64 * 576 + 4 * 32 * 256;
  const dim3 tg((FC_IN + 31) / 32, (L.H + 31) / 32);
  if (prec == ALEPPO_BF16) {
    hipLaunchKernelGGL(pack_conv_dgrad_kernel<bf16>, dim3((n + 255) / 256), dim3(256), 0, s, P + L.off[P_W3],
                       P + L.off[P_W2], static_cast<bf16 *>(W3d), static_cast<bf16 *>(W2d));
    hipLaunchKernelGGL(transpose_cast_kernel<bf16>, tg, dim3(256), 0, s, P + L.off[P_WFC], static_cast<bf16 *>(WfcT),
                       L.H, FC_IN);
  } else {
    hipLaunchKernelGGL(pack_conv_dgrad_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, P + L.off[P_W3],
                       P + L.off[P_W2], static_cast<float *>(W3d), static_cast<float *>(W2d));
    hipLaunchKernelGGL(transpose_cast_kernel<float>, tg, dim3(256), 0, s, P + L.off[P_WFC],
                       static_cast<float *>(WfcT), L.H, FC_IN);
  }
}

// masked sums of the per-sample metric arrays (log_data's masked means, train.cc:163-210): block per (epoch, mb)
__global__ __launch_bounds__(256) void metrics_reduce_kernel(const float *ps, size_t field_stride, const uint8_t *mask_n,
                                                              long B, int M, float *out) {
  __shared__ float s4[4];
  const int mi = blockIdx.x, mb = mi % M;
  const uint8_t *m = mask_n + (size_t)mb * B;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, cnt = 0.f;
  for (long i = threadIdx.x; i < B; i += 256)
    if (m[i]) {
      cnt += 1.f;
#pragma unroll
      for (int f = 0; f < 5; ++f)
        acc[f] += ps[f * field_stride + (size_t)mi * B + i];
    }
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    const float v = block_sum_256(acc[f], s4);
    if (threadIdx.x == 0)
      out[mi * 8 + f] = v;
  }
  cnt = block_sum_256(cnt, s4);
  if (threadIdx.x == 0)
    out[mi * 8 + 5] = cnt;
}
void launch_metrics_reduce(hipStream_t s, const float *ps, size_t field_stride, const uint8_t *mask_n, long B, int M,
                           int epochs, float *out) {
  hipLaunchKernelGGL(metrics_reduce_kernel, dim3(epochs * M), dim3(256), 0, s, ps, field_stride, mask_n, B, M, out);
}

// ================================================================================================
// Boundary-only layout conversions (parity dumps / caller-supplied batches; not on the hot path).
// ================================================================================================
__global__ void obs_unpack_kernel(const uint32_t *obs, uint8_t *out, SampleMap map) { // -> NCHW u8 [n][4][7056]
  const long n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= FRAME_PIX)
    return;
  const uint32_t w = obs[(n / map.TP) * map.s1 + (n % map.TP) * map.s0 + map.base + i];
  uint8_t *o = out + (size_t)n * 4 * FRAME_PIX + i;
  o[0] = w & 255u;
  o[FRAME_PIX] = (w >> 8) & 255u;
  o[2 * FRAME_PIX] = (w >> 16) & 255u;
  o[3 * FRAME_PIX] = w >> 24;
}
__global__ void obs_pack_kernel(const uint8_t *in, uint32_t *obs, SampleMap map) {
  const long n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= FRAME_PIX)
    return;
  const uint8_t *p = in + (size_t)n * 4 * FRAME_PIX + i;
  obs[(n / map.TP) * map.s1 + (n % map.TP) * map.s0 + map.base + i] =
      (uint32_t)p[0] | ((uint32_t)p[FRAME_PIX] << 8) | ((uint32_t)p[2 * FRAME_PIX] << 16) |
      ((uint32_t)p[3 * FRAME_PIX] << 24);
}
void launch_obs_unpack(hipStream_t s, const uint32_t *obs, uint8_t *out, long nsamp, SampleMap map) {
  hipLaunchKernelGGL(obs_unpack_kernel, dim3((FRAME_PIX + 255) / 256, (unsigned)nsamp), dim3(256), 0, s, obs, out, map);
}
void launch_obs_pack(hipStream_t s, const uint8_t *in, uint32_t *obs, long nsamp, SampleMap map) {
  hipLaunchKernelGGL(obs_pack_kernel, dim3((FRAME_PIX + 255) / 256, (unsigned)nsamp), dim3(256), 0, s, in, obs, map);
}

// [T][E][inner] (elements of `elem` bytes, row pitch src_pitch bytes per t) -> [E][T][inner]
__global__ void transpose_tm_kernel(const uint8_t *src, size_t src_pitch, uint8_t *dst, int E, int T, int inner,
                                    int elem) {
  const long total = (long)E * T * inner;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % inner);
    const long et = i / inner;
    const int t = (int)(et % T), e = (int)(et / T);
    const uint8_t *s = src + (size_t)t * src_pitch + ((size_t)e * inner + k) * elem;
    uint8_t *d = dst + (size_t)i * elem;
    for (int b = 0; b < elem; ++b)
      d[b] = s[b];
  }
}
void launch_transpose_tm_pitched(hipStream_t s, const void *src_tm, size_t pitch, void *dst_em, int E, int T, int inner,
                                 int elem) {
  hipLaunchKernelGGL(transpose_tm_kernel, dim3(256), dim3(256), 0, s, static_cast<const uint8_t *>(src_tm), pitch,
                     static_cast<uint8_t *>(dst_em), E, T, inner, elem);
}

// action/value heads only (aleppo_forward): one wave per row
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float *__restrict__ h, const float *__restrict__ Wh,
                                                         const float *__restrict__ bh, float *logits, float *values,
                                                         long n, int H, int A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= n)
    return;
  for (int a = 0; a <= A; ++a) {
    float s = 0.f;
    for (int j = lane; j < H; j += 64)
      s += h[(size_t)r * H + j] * Wh[(size_t)a * H + j];
    s = wave_sum(s);
    if (lane == 0) {
      if (a < A)
        logits[r * A + a] = s + bh[a];
      else
        values[r] = s + bh[a];
    }
  }
}
void launch_heads_fwd(hipStream_t s, const float *h, const float *Wh, const float *bh, float *logits, float *values,
                      long n, int H, int A) {
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, h, Wh, bh, logits, values, n, H,
                     A);
}

__global__ void logsoftmax_rows_kernel(const float *in, float *out, long rows, int A) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows)
    return;
  const float *z = in + r * A;
  float mx = z[0];
  for (int k = 1; k < A; ++k)
    mx = fmaxf(mx, z[k]);
  float s = 0.f;
  for (int k = 0; k < A; ++k)
    s += expf(z[k] - mx);
  const float lse = mx + logf(s);
  for (int k = 0; k < A; ++k)
    out[r * A + k] = z[k] - lse;
}
void launch_logsoftmax_rows(hipStream_t s, const float *in, float *out, long rows, int A) {
  hipLaunchKernelGGL(logsoftmax_rows_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, in, out, rows, A);
}

// ================================================================================================
// Stateless operators of the reference's free functions (parity tests).
// ================================================================================================
// vision.cc:8-32 interpolate(mode=area) 210x160 -> 84x84, float in / float out
__global__ void area_resize_kernel(const float *in, float *out) {
  const long n = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= FRAME_PIX)
    return;
  const int i = pix / 84, j = pix - i * 84;
  const int y0 = (i * RAW_H) / 84, y1 = ((i + 1) * RAW_H + 83) / 84, x0 = (j * RAW_W) / 84,
            x1 = ((j + 1) * RAW_W + 83) / 84;
  const float *src = in + (size_t)n * RAW_H * RAW_W;
  float s = 0.f;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x)
      s += src[y * RAW_W + x];
  out[(size_t)n * FRAME_PIX + pix] = s / (float)((y1 - y0) * (x1 - x0));
}
void launch_area_resize(hipStream_t s, const float *in, float *out, long n) {
  hipLaunchKernelGGL(area_resize_kernel, dim3((FRAME_PIX + 255) / 256, (unsigned)n), dim3(256), 0, s, in, out);
}
// vision.cc:51,71-84
__global__ void rgb_to_gray_kernel(const float *in, float *out) {
  const long n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= FRAME_PIX)
    return;
  const float *s = in + (size_t)n * 3 * FRAME_PIX + p;
  {
#pragma clang fp contract(off)
    const float a = s[0] * 0.2125f, b = s[FRAME_PIX] * 0.7154f, c = s[2 * FRAME_PIX] * 0.0721f;
    out[(size_t)n * FRAME_PIX + p] = (a + b) + c;
  }
}
void launch_rgb_to_gray(hipStream_t s, const float *in, float *out, long n) {
  hipLaunchKernelGGL(rgb_to_gray_kernel, dim3((FRAME_PIX + 255) / 256, (unsigned)n), dim3(256), 0, s, in, out);
}
} // namespace aleppo
