// api.hip - C ABI (include/aleppo.h) over the HIP kernels: context, rollout protocol, PPO update,
// RCCL gradient all-reduce, parity read-back.  Host-side orchestration only; every number is produced
// by a HIP kernel on the device - there is no CPU fallback (a missing device is ALEPPO_ERR_NO_DEVICE).
#include "common.hpp"
#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <cstring>
#include <rccl/rccl.h>

using namespace aleppo;

namespace aleppo {

static thread_local std::string g_err;
int set_err(Ctx *c, int code, const std::string &msg) {
  if (c)
    c->err = msg;
  g_err = msg;
  return code;
}

// ------------------------------------------------------------------ per-context kernel-selection switches
static thread_local const Tuning *g_tuning = nullptr;
Tuning tuning_from_env() { // read once per context, in aleppo_create
  auto flag = [](const char *name, bool dflt) {
    const char *e = std::getenv(name);
    return e ? std::atoi(e) != 0 : dflt;
  };
  Tuning t;
  t.patch_conv = !flag("ALEPPO_GENERIC_CONV", false);
  t.fc_pipe = flag("ALEPPO_FC_PIPE", true);
  t.fused_fwd = flag("ALEPPO_FWD_FUSED", true);
  if (const char *e = std::getenv("ALEPPO_BWD_FUSED"))
    t.fused_bwd = std::atoi(e);
  if (const char *e = std::getenv("ALEPPO_FUSED_ACT"))
    t.fused_act = std::atoi(e);
  return t;
}
const Tuning &tuning() {
  static const Tuning dflt = tuning_from_env();
  return g_tuning ? *g_tuning : dflt;
}
void set_tuning(const Tuning *t) { g_tuning = t; }

// ------------------------------------------------------------------ parameter layout
static inline size_t align64(size_t x) { return (x + 63) / 64 * 64; }
void ParamLayout::init(int H_, int A_) {
  H = H_;
  A = A_;
  size[P_WH] = (size_t)(A + 1) * H;
  size[P_BH] = (size_t)A + 1;
  size[P_WFC] = (size_t)H * FC_IN;
  size[P_BFC] = (size_t)H;
  size[P_W3] = 64 * 576;
  size[P_B3] = 64;
  size[P_W2] = 64 * 512;
  size[P_B2] = 64;
  size[P_W1] = 32 * 256;
  size[P_B1] = 32;
  off[0] = 0;
  for (int i = 0; i < P_COUNT; ++i)
    off[i + 1] = off[i] + align64(size[i]);
  bucket0_end = off[P_W3];
}
size_t ParamLayout::reference_count() const {
  size_t n = 0;
  for (int i = 0; i < P_COUNT; ++i)
    n += size[i];
  return n;
}

// libtorch parameters() order: c1w[32,4,8,8] c1b c2w[64,32,4,4] c2b c3w[64,64,3,3] c3b fcw[H,3136] fcb aw[A,H] ab
// vw[1,H] vb.  Internal: conv weights [oc][(kh,kw,c)]; fc weight columns in (y,x,c) order (NHWC flatten) instead
// of libtorch's (c,y,x); heads stacked [A+1][H] (value head last).
template <bool TO_INTERNAL> static void permute_params(const ParamLayout &L, const float *src, float *dst) {
  const int H = L.H, A = L.A;
  size_t r = 0; // running reference offset
  auto conv = [&](int OC, int C, int KH, int KW, ParamId wid, ParamId bid) {
    const size_t wn = (size_t)OC * C * KH * KW;
    for (int oc = 0; oc < OC; ++oc)
      for (int c = 0; c < C; ++c)
        for (int kh = 0; kh < KH; ++kh)
          for (int kw = 0; kw < KW; ++kw) {
            const size_t ri = r + (((size_t)oc * C + c) * KH + kh) * KW + kw;
            const size_t ii = L.off[wid] + (size_t)oc * (KH * KW * C) + (size_t)(kh * KW + kw) * C + c;
            if (TO_INTERNAL)
              dst[ii] = src[ri];
            else
              dst[ri] = src[ii];
          }
    r += wn;
    for (int oc = 0; oc < OC; ++oc) {
      if (TO_INTERNAL)
        dst[L.off[bid] + oc] = src[r + oc];
      else
        dst[r + oc] = src[L.off[bid] + oc];
    }
    r += OC;
  };
  conv(32, 4, 8, 8, P_W1, P_B1);
  conv(64, 32, 4, 4, P_W2, P_B2);
  conv(64, 64, 3, 3, P_W3, P_B3);
  for (int o = 0; o < H; ++o)
    for (int c = 0; c < 64; ++c)
      for (int p = 0; p < 49; ++p) {
        const size_t ri = r + (size_t)o * FC_IN + c * 49 + p, ii = L.off[P_WFC] + (size_t)o * FC_IN + p * 64 + c;
        if (TO_INTERNAL)
          dst[ii] = src[ri];
        else
          dst[ri] = src[ii];
      }
  r += (size_t)H * FC_IN;
  auto lin = [&](size_t n, size_t ioff) {
    for (size_t i = 0; i < n; ++i) {
      if (TO_INTERNAL)
        dst[ioff + i] = src[r + i];
      else
        dst[r + i] = src[ioff + i];
    }
    r += n;
  };
  lin(H, L.off[P_BFC]);
  lin((size_t)A * H, L.off[P_WH]);          // action_head.weight
  lin(A, L.off[P_BH]);                      // action_head.bias
  lin(H, L.off[P_WH] + (size_t)A * H);      // value_head.weight
  lin(1, L.off[P_BH] + A);                  // value_head.bias
}
void params_to_internal(const ParamLayout &L, const float *ref, float *internal) {
  std::fill(internal, internal + L.total(), 0.0f);
  permute_params<true>(L, ref, internal);
}
void params_to_reference(const ParamLayout &L, const float *internal, float *ref) {
  permute_params<false>(L, internal, ref);
}

} // namespace aleppo

// ------------------------------------------------------------------ helpers
// every stateful entry point: the caller's thread may have another device current, and the kernel-selection switches
// are this context's
#define CHECK_CTX_ANY(c)                                                                                               \
  do {                                                                                                                 \
    if (!(c))                                                                                                          \
      return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "null context");                                            \
    if (hipSetDevice((c)->cfg.device_ordinal) != hipSuccess)                                                           \
      return set_err(const_cast<aleppo_ctx *>(c), ALEPPO_ERR_HIP, "hipSetDevice failed");                              \
    set_tuning(&(c)->tune);                                                                                            \
    if ((c)->failed)                                                                                                   \
      return set_err(const_cast<aleppo_ctx *>(c), ALEPPO_ERR_RUNTIME, (c)->fail_msg);                                  \
  } while (0)
// Between aleppo_arm_step and aleppo_release_step the stream is parked on the release word: anything that enqueues behind
// it and then waits (or rewrites what the parked kernels read) would dead-lock, so every entry point but the release
// refuses.
#define CHECK_CTX(c)                                                                                                   \
  do {                                                                                                                 \
    CHECK_CTX_ANY(c);                                                                                                  \
    if ((c)->armed)                                                                                                    \
      return set_err(const_cast<aleppo_ctx *>(c), ALEPPO_ERR_RUNTIME,                                                  \
                     "a step is armed: call aleppo_release_step first");                                               \
  } while (0)
#define CHECK_ASYNC(c)                                                                                                 \
  do {                                                                                                                 \
    if ((c)->async_err != hipSuccess) {                                                                                \
      const hipError_t e_ = (c)->async_err;                                                                            \
      (c)->async_err = hipSuccess;                                                                                     \
      return set_err((c), ALEPPO_ERR_HIP, std::string("asynchronous HIP failure: ") + hipGetErrorString(e_));          \
    }                                                                                                                  \
  } while (0)
#define NCCLCHK(c, x)                                                                                                  \
  do {                                                                                                                 \
    ncclResult_t r_ = (x);                                                                                             \
    if (r_ != ncclSuccess)                                                                                             \
      return set_err((c), ALEPPO_ERR_HIP, std::string(#x) + ": " + ncclGetErrorString(r_));                            \
  } while (0)

// A failure after which the rollout / learner state is undefined: the context refuses every later call (sticky), any
// gate still on the stream is released so that the stream drains, and nothing is freed or reused before aleppo_destroy
// (kernels that are still queued may read the caller's frame buffers until then).
static int fail_ctx(Ctx *c, int code, const std::string &msg) {
  c->failed = true;
  c->fail_msg = msg + " - the context is unusable: destroy it";
  c->armed = false;
  c->act_queued_slot = -1;
  if (c->h_go)
    __atomic_store_n(c->h_go, ~0ull, __ATOMIC_RELEASE);
  return set_err(c, code, c->fail_msg);
}

static size_t tsz(const Ctx *c) { return c->prec == ALEPPO_BF16 ? 2 : 4; }
// A context only ever touches its OWN streams after aleppo_create: no null-stream operation, no hipFree, no
// hipDeviceSynchronize.  Those calls wait for (hipFree / hipHostFree / hipDeviceSynchronize) or are ordered against
// (null stream) other streams of the device - and another context's stream may be parked behind its release word,
// which only ITS owner thread lifts (the cause of the two-context hang of round 2).
template <class T> static hipError_t dalloc(T **p, size_t bytes, hipStream_t st) {
  hipError_t e = hipMalloc(reinterpret_cast<void **>(p), bytes ? bytes : 16);
  if (e == hipSuccess)
    e = hipMemsetAsync(*p, 0, bytes ? bytes : 16, st);
  // the fill is asynchronous: wait for it, or a kernel on another stream of the context could use the buffer first and
  // have its results wiped afterwards (round 2: a late fill of the ticket counter / the metric planes)
  if (e == hipSuccess)
    e = hipStreamSynchronize(st);
  return e;
}
// host <-> device copy on the context's main stream, complete when the call returns
static hipError_t copy_sync(Ctx *c, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
  return e == hipSuccess ? hipStreamSynchronize(c->stream) : e;
}
// device / pinned-host memory a context has outgrown: kept until aleppo_destroy (see above)
static void retire(Ctx *c, void *dev) {
  if (dev)
    c->retired.push_back(dev);
}
static void retire_host(Ctx *c, void *host) {
  if (host)
    c->retired_host.push_back(host);
}

// a failure inside a helper that cannot return a status is kept in the context and reported by CHECK_ASYNC at the end of
// the entry point (never dropped)
static inline void note(Ctx *c, hipError_t e) {
  if (e != hipSuccess && c->async_err == hipSuccess)
    c->async_err = e;
}
static void prof_begin(Ctx *c, int cls, hipStream_t st = nullptr) {
  if (!c->prof_on)
    return;
  if (!st)
    st = c->stream;
  ProfClass &p = c->prof[cls];
  if (p.used == p.start.size()) {
    hipEvent_t a = nullptr, b = nullptr;
    note(c, hipEventCreate(&a));
    note(c, hipEventCreate(&b));
    p.start.push_back(a);
    p.stop.push_back(b);
  }
  note(c, hipEventRecord(p.start[p.used], st));
}
static void prof_end(Ctx *c, int cls, hipStream_t st = nullptr) { // same stream as the matching prof_begin
  if (!c->prof_on)
    return;
  if (!st)
    st = c->stream;
  ProfClass &p = c->prof[cls];
  note(c, hipEventRecord(p.stop[p.used], st));
  p.used++;
}

static SampleMap train_map(const Ctx *c, long n0) {
  // sample n = e*T + t lives in slot (e, t) of obs [E][T+1][7056]
  return SampleMap{c->T, (long)(c->T + 1) * FRAME_PIX, (long)FRAME_PIX, 0, (int)n0};
}
static SampleMap slot_map(const Ctx *c, int t) {
  return SampleMap{1, (long)(c->T + 1) * FRAME_PIX, 0, (long)t * FRAME_PIX, 0};
}

static const float *Pf(const Ctx *c, ParamId id) { return c->P + c->L.off[id]; }
// element i of a rollout plane stored as RT (float or half)
static void *rp(const Ctx *c, void *plane, size_t i) { return static_cast<char *>(plane) + i * c->rsz; }
static const void *Pcw(const Ctx *c, ParamId id) {
  return static_cast<const char *>(c->Pc) + c->L.off[id] * tsz(c);
}

// conv stack forward for ns samples addressed by map -> c->h
// returns the number of split-K partial slabs of h (1 unless max_parts allows the pipelined fc kernel to split)
static int net_forward(Ctx *c, const uint32_t *obs, SampleMap map, long ns, int max_parts = 1) {
  if (c->prec == ALEPPO_BF16 && use_patch_kernels() && tuning().fused_fwd) { // one launch: a1 / a2 are only written
    prof_begin(c, ALEPPO_K_CONV_FWD);
    patch_fwd_fused(c->stream, obs, map, Pcw(c, P_W1), Pf(c, P_B1), Pcw(c, P_W2), Pf(c, P_B2), Pcw(c, P_W3), Pf(c, P_B3),
                    c->a1, c->a2, c->a3, ns);
    prof_end(c, ALEPPO_K_CONV_FWD);
    prof_begin(c, ALEPPO_K_FC_FWD);
    const int parts = fc_fwd(c->stream, c->prec, c->a3, Pcw(c, P_WFC), Pf(c, P_BFC), c->h, ns, c->H, max_parts);
    prof_end(c, ALEPPO_K_FC_FWD);
    return parts;
  }
  prof_begin(c, ALEPPO_K_CONV1_FWD);
  conv1_fwd(c->stream, c->prec, obs, map, Pcw(c, P_W1), Pf(c, P_B1), c->a1, ns);
  prof_end(c, ALEPPO_K_CONV1_FWD);
  prof_begin(c, ALEPPO_K_CONV2_FWD);
  conv2_fwd(c->stream, c->prec, c->a1, Pcw(c, P_W2), Pf(c, P_B2), c->a2, ns);
  prof_end(c, ALEPPO_K_CONV2_FWD);
  prof_begin(c, ALEPPO_K_CONV3_FWD);
  conv3_fwd(c->stream, c->prec, c->a2, Pcw(c, P_W3), Pf(c, P_B3), c->a3, ns);
  prof_end(c, ALEPPO_K_CONV3_FWD);
  prof_begin(c, ALEPPO_K_FC_FWD);
  const int parts = fc_fwd(c->stream, c->prec, c->a3, Pcw(c, P_WFC), Pf(c, P_BFC), c->h, ns, c->H, max_parts);
  prof_end(c, ALEPPO_K_FC_FWD);
  return parts;
}

static void refresh_compute_copies(Ctx *c) {
  if (c->prec == ALEPPO_BF16)
    launch_cast_params(c->stream, c->P, c->Pc, (long)c->L.total());
  launch_pack_dgrad(c->stream, c->P, c->L, c->W2d, c->W3d, c->WfcT, c->prec);
}

// ------------------------------------------------------------------ lifetime
extern "C" int aleppo_abi_version(void) { return ALEPPO_ABI_VERSION; }
extern "C" const char *aleppo_last_error(const aleppo_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

static int select_device(int ordinal) {
  set_tuning(nullptr); // (the stateless operators and aleppo_create run on the process defaults)
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return set_err(nullptr, ALEPPO_ERR_NO_DEVICE,
                   "no HIP device visible: libaleppo has no CPU fallback (needs an MI355X / gfx950)");
  if (ordinal < 0 || ordinal >= n)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "device_ordinal out of range");
  if (hipSetDevice(ordinal) != hipSuccess)
    return set_err(nullptr, ALEPPO_ERR_HIP, "hipSetDevice failed");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess)
    return set_err(nullptr, ALEPPO_ERR_HIP, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_err(nullptr, ALEPPO_ERR_NO_DEVICE,
                   std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  return ALEPPO_OK;
}

extern "C" int aleppo_device_check(int device_ordinal) { return select_device(device_ordinal); }

extern "C" int aleppo_create(const aleppo_config *cfg, aleppo_ctx **out) {
  if (!cfg || !out)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (cfg->abi_version != ALEPPO_ABI_VERSION)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "abi_version mismatch");
  // same checks as Rollout::Rollout (rollout.cc:48-59) where they apply
  if (cfg->num_envs <= 0)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "Total environments must be greater than 0.");
  if (cfg->num_envs > MAX_ENVS_PER_RANK)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "num_envs per rank must be <= 8192");
  if (cfg->horizon <= 0)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "Horizon must be greater than 0.");
  if (cfg->frame_stack != 4)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "frame_stack must be 4 (conv1 has 4 input channels)");
  if (cfg->num_actions < 1 || cfg->num_actions > MAX_ACTIONS)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "num_actions must be in [1,18]");
  if (cfg->hidden_size < 32 || cfg->hidden_size > 512 || cfg->hidden_size % 32)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "hidden_size must be a multiple of 32 in [32,512]");
  if (cfg->precision != ALEPPO_FP32 && cfg->precision != ALEPPO_BF16)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "precision must be ALEPPO_FP32 or ALEPPO_BF16");
  if (cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "bad world_size / rank");
  if (cfg->rollout_precision != ALEPPO_ROLLOUT_FP32 && cfg->rollout_precision != ALEPPO_ROLLOUT_FP16)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "rollout_precision must be ALEPPO_ROLLOUT_FP32 or _FP16");
  int rc = select_device(cfg->device_ordinal);
  if (rc)
    return rc;

  aleppo_ctx *c = new aleppo_ctx();
  c->cfg = *cfg;
  c->tune = tuning_from_env();
  set_tuning(&c->tune);
  if (c->cfg.adam_beta1 == 0.f)
    c->cfg.adam_beta1 = 0.9f;
  if (c->cfg.adam_beta2 == 0.f)
    c->cfg.adam_beta2 = 0.999f;
  if (c->cfg.adam_eps == 0.f)
    c->cfg.adam_eps = 1e-5f;
  c->E = cfg->num_envs;
  c->T = cfg->horizon;
  c->A = cfg->num_actions;
  c->H = cfg->hidden_size;
  c->prec = cfg->precision;
  c->world = cfg->world_size;
  c->rank = cfg->rank;
  c->N = (long)c->E * c->T;
  c->rt16 = cfg->rollout_precision == ALEPPO_ROLLOUT_FP16;
  c->rsz = c->rt16 ? 2 : 4;
  c->maxB = cfg->max_minibatch > 0 ? std::max<long>(cfg->max_minibatch, c->E) : std::max<long>(c->N, c->E);
  c->L.init(c->H, c->A);
  const int E = c->E, T = c->T, A = c->A, H = c->H;
  const size_t ts = tsz(c), PT = c->L.total();
#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      set_err(nullptr, ALEPPO_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                               \
      aleppo_destroy(c);                                                                                               \
      return ALEPPO_ERR_HIP;                                                                                           \
    }                                                                                                                  \
  } while (0)
  CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  // (the update's two side streams are created by the first aleppo_train: a context that only acts holds ONE stream.
  // The runtime multiplexes streams onto GPU_MAX_HW_QUEUES - default 4 - hardware queues, and a kernel that lands in the
  // queue of another context's parked stream waits behind its gate: tests/tools/parkprobe.hip)
  CK(hipEventCreateWithFlags(&c->ev_bucket0, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&c->ev_comm0, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&c->ev_comm1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&c->ev_tmp, hipEventDisableTiming));
  for (hipEvent_t *e : {&c->ev_head, &c->ev_dz3, &c->ev_dz2, &c->ev_wg})
    CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&c->ev_adam, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
  CK(dalloc(&c->obs, (size_t)E * (T + 1) * FRAME_PIX * 4, c->stream));
  c->step_rec_bytes = ((size_t)7 * E + 15) / 16 * 16;
  CK(dalloc(&c->step_rec, c->step_rec_bytes * T, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->values_tm), (size_t)(T + 1) * E * c->rsz, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->logits_tm), (size_t)(T + 1) * E * A * c->rsz, c->stream));
  CK(dalloc(&c->actions_tm, (size_t)(T + 1) * E * 4, c->stream));
  CK(dalloc(&c->lut, 256, c->stream));
  CK(dalloc(&c->d_start, (size_t)E, c->stream));
  CK(dalloc(&c->d_frames, (size_t)E * 2 * RAW_H * RAW_W, c->stream));
  CK(dalloc(&c->d_noise, (size_t)2 * E * A * 4, c->stream));
  CK(dalloc(&c->d_err, 16, c->stream));
  CK(dalloc(&c->d_done, 16, c->stream));
  {
    uint8_t ident[256];
    for (int i = 0; i < 256; ++i)
      ident[i] = (uint8_t)i;
    CK(copy_sync(c, c->lut, ident, 256, hipMemcpyHostToDevice));
  }
  CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_actions), (size_t)(E + 8) * 8, hipHostMallocMapped));
  CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_step), c->step_rec_bytes + E, hipHostMallocDefault));
  CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_rec), c->step_rec_bytes * T, hipHostMallocDefault));
  std::memset(c->h_rec, 0, c->step_rec_bytes * T);
  CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_frames), (size_t)E * 2 * RAW_H * RAW_W, hipHostMallocDefault));
  CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_noise), (size_t)2 * E * A * 4, hipHostMallocDefault));
  CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_go), 64, hipHostMallocMapped));
  std::memset(c->h_go, 0, 64);
  {
    // exit condition of the slot-ahead gate (gate_kernel): an emulator step takes milliseconds, so two minutes mean the
    // host is gone.  ALEPPO_GATE_TIMEOUT_MS / ALEPPO_OPT_GATE_TIMEOUT_MS change it (the tests use 100 ms).
    const char *e = std::getenv("ALEPPO_GATE_TIMEOUT_MS");
    const double ms = e ? std::max(1.0, std::atof(e)) : 120000.0;
    c->gate_timeout_ticks = (unsigned long long)(ms * 1e5); // 100 MHz wall clock
  }
  CK(hipHostMalloc(reinterpret_cast<void **>(&c->h_err), 16, hipHostMallocDefault));
  std::memset(c->h_actions, 0, (size_t)(E + 8) * 8);
  CK(dalloc(reinterpret_cast<char **>(&c->adv_n), (size_t)c->N * c->rsz, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->ret_n), (size_t)c->N * c->rsz, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->oldlp_n), (size_t)c->N * A * c->rsz, c->stream));
  CK(dalloc(&c->act_n, (size_t)c->N * 4, c->stream));
  CK(dalloc(&c->mask_n, (size_t)c->N, c->stream));
  CK(dalloc(&c->mask_counts, 4096 * 4, c->stream));
  CK(dalloc(&c->P, PT * 4, c->stream));
  CK(dalloc(&c->G, PT * 4, c->stream));
  CK(dalloc(&c->M1, PT * 4, c->stream));
  CK(dalloc(&c->M2, PT * 4, c->stream));
  if (c->prec == ALEPPO_BF16)
    CK(dalloc(reinterpret_cast<char **>(&c->Pc), PT * 2, c->stream));
  else
    c->Pc = c->P;
  CK(dalloc(reinterpret_cast<char **>(&c->W2d), (size_t)4 * 32 * 256 * ts, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->W3d), (size_t)64 * 576 * ts, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->WfcT), (size_t)FC_IN * H * ts, c->stream));
  const size_t mb = (size_t)c->maxB;
  CK(dalloc(reinterpret_cast<char **>(&c->a1), mb * A1_PIX * A1_C * ts, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->a2), mb * A2_PIX * A2_C * ts, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->a3), mb * FC_IN * ts, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->dz1), mb * A1_PIX * A1_C * ts, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->dz2), mb * A2_PIX * A2_C * ts, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->dz3), mb * FC_IN * ts, c->stream));
  CK(dalloc(&c->h, (size_t)FC_FWD_MAX_PARTS * mb * H * 4, c->stream)); // up to FC_FWD_MAX_PARTS split-K slabs
  CK(dalloc(&c->hpart, (size_t)FC_SPLITS * E * H * 4, c->stream));
  CK(dalloc(reinterpret_cast<char **>(&c->dh), mb * H * ts, c->stream));
  CK(dalloc(&c->logits_b, mb * A * 4, c->stream));
  CK(dalloc(&c->values_b, mb * 4, c->stream));
  // slabs: [W1|b1|W2|b2|W3|b3|Wfc|bfc|Wh|bh]
  c->slab_off[0] = 0;
  const size_t sl[10] = {(size_t)MAXS_C1 * 32 * 256,       (size_t)MAXS_C1 * 32, (size_t)MAXS_C2 * 64 * 512,
                         (size_t)MAXS_C2 * 64,              (size_t)MAXS_C3 * 64 * 576, (size_t)MAXS_C3 * 64,
                         (size_t)MAXS_FC * H * FC_IN,       (size_t)MAXS_HEAD * H, (size_t)MAXS_HEAD * (A + 1) * H,
                         (size_t)MAXS_HEAD * (A + 1)};
  for (int i = 0; i < 10; ++i)
    c->slab_off[i + 1] = c->slab_off[i] + align64(sl[i]);
  c->slab_floats = c->slab_off[10];
  CK(dalloc(&c->slab, c->slab_floats * 4, c->stream));
  CK(dalloc(&c->sumsq_part, 1024 * 4, c->stream));
  CK(dalloc(&c->adv_stats, 64, c->stream));
  CK(hipStreamSynchronize(c->stream)); // (own streams only: see dalloc)
#undef CK
  *out = c;
  return ALEPPO_OK;
}

extern "C" void aleppo_destroy(aleppo_ctx *c) {
  if (!c)
    return;
  set_tuning(nullptr);
  (void)hipSetDevice(c->cfg.device_ordinal);
  if (c->h_go) // an armed step parks the stream on the release word: let it go before waiting for it
    __atomic_store_n(c->h_go, ~0ull, __ATOMIC_RELEASE);
  for (hipStream_t st : {c->wg_stream, c->comm_stream, c->stream})
    if (st)
      hipStreamSynchronize(st);
  // From here on hipFree / hipHostFree: each waits for every stream of the device.  If another context of this process
  // has a step armed on another thread, that wait lasts until its owner releases it (never call aleppo_destroy from the
  // thread that owns an armed context: INTEGRATION.md, threading).
  if (c->graph_exec)
    hipGraphExecDestroy(c->graph_exec);
  if (c->graph)
    hipGraphDestroy(c->graph);
  if (c->nccl_comm)
    ncclCommDestroy(static_cast<ncclComm_t>(c->nccl_comm));
  void *dev[] = {c->obs,   c->step_rec, c->values_tm, c->logits_tm, c->actions_tm, c->lut,     c->d_start,
                 c->d_frames, c->d_noise, c->d_err,  c->d_done, c->adv_n,     c->ret_n,      c->oldlp_n, c->act_n,
                 c->mask_n, c->mask_counts, c->P,    c->G,         c->Gs,         c->M1,      c->M2,
                 c->W2d,   c->W3d,      c->WfcT,      c->a1,        c->a2,         c->a3,      c->dz1,
                 c->dz2,   c->dz3,      c->h,         c->hpart,     c->dh,        c->logits_b,   c->values_b, c->slab,
                 c->sumsq_part, c->metric_ps, c->metric_red, c->grad_norms, c->adv_stats, c->stage_u8, c->stage_obs,
                 c->adam_sched, c->rb_tmp[0], c->rb_tmp[1]};
  for (void *p : dev)
    if (p)
      hipFree(p);
  for (void *p : c->retired)
    hipFree(p);
  if (c->Pc && c->Pc != c->P)
    hipFree(c->Pc);
  void *host[] = {c->h_go, c->h_actions, c->h_step, c->h_rec, c->h_frames, c->h_noise, c->h_err, c->h_metric_red,
                  c->h_adam_sched};
  for (void *p : host)
    if (p)
      hipHostFree(p);
  for (void *p : c->retired_host)
    hipHostFree(p);
  for (auto &pc : c->prof)
    for (size_t i = 0; i < pc.start.size(); ++i) {
      hipEventDestroy(pc.start[i]);
      hipEventDestroy(pc.stop[i]);
    }
  for (hipEvent_t e : {c->ev_bucket0, c->ev_comm0, c->ev_comm1, c->ev_tmp, c->ev_adam, c->ev_pack, c->ev_head, c->ev_dz3,
                       c->ev_dz2, c->ev_wg})
    if (e)
      hipEventDestroy(e);
  if (c->stream)
    hipStreamDestroy(c->stream);
  if (c->comm_stream)
    hipStreamDestroy(c->comm_stream);
  if (c->wg_stream)
    hipStreamDestroy(c->wg_stream);
  delete c;
}

extern "C" int aleppo_synchronize(aleppo_ctx *c) {
  CHECK_CTX(c);
  for (hipStream_t st : {c->wg_stream, c->comm_stream, c->stream})
    if (st)
      HIPCHK(c, hipStreamSynchronize(st));
  return ALEPPO_OK;
}

// ------------------------------------------------------------------ parameters
extern "C" int aleppo_param_count(const aleppo_ctx *c, size_t *count) {
  CHECK_CTX(c);
  if (!count)
    return ALEPPO_ERR_INVALID_ARGUMENT;
  *count = c->L.reference_count();
  return ALEPPO_OK;
}
extern "C" int aleppo_load_params(aleppo_ctx *c, const float *flat, size_t count) {
  CHECK_CTX(c);
  if (!flat || count != c->L.reference_count())
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "load_params: wrong element count");
  std::vector<float> tmp(c->L.total());
  params_to_internal(c->L, flat, tmp.data());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemsetAsync(c->M1, 0, tmp.size() * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->M2, 0, tmp.size() * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->G, 0, tmp.size() * 4, c->stream));
  HIPCHK(c, copy_sync(c, c->P, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice)); // (own stream only: see dalloc)
  c->adam_step = 0;
  c->pre_acted = -1;
  refresh_compute_copies(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ALEPPO_OK;
}
static int export_flat(aleppo_ctx *c, const float *dev, float *flat, size_t count) {
  if (!flat || count != c->L.reference_count())
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "export: wrong element count");
  std::vector<float> tmp(c->L.total());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_sync(c, tmp.data(), dev, tmp.size() * 4, hipMemcpyDeviceToHost));
  params_to_reference(c->L, tmp.data(), flat);
  return ALEPPO_OK;
}
extern "C" int aleppo_export_params(aleppo_ctx *c, float *flat, size_t count) {
  CHECK_CTX(c);
  return export_flat(c, c->P, flat, count);
}
extern "C" int aleppo_export_grads(aleppo_ctx *c, float *flat, size_t count) {
  CHECK_CTX(c);
  // clip_grad_norm_ scales in place (train.cc:42-44); the Adam kernel applies the same factor on the fly and leaves G as the
  // backward pass produced it (a separate scaled copy cost 4 bytes per parameter and optimizer step).  The factor is one
  // fp32 product of the stored pre-clip norm: applied here it gives the bits the kernel multiplied into its update.
  const int rc = export_flat(c, c->G, flat, count);
  if (rc || c->last_epochs * c->last_M == 0)
    return rc;
  const size_t nm = (size_t)c->last_epochs * c->last_M;
  const float norm = c->h_metric_red[nm * 8 + nm - 1];
  float coef = c->cfg.max_gradient_norm / (norm + 1e-6f);
  coef = std::fmin(coef, 1.0f);
  for (size_t i = 0; i < count; ++i)
    flat[i] *= coef;
  return ALEPPO_OK;
}

extern "C" int aleppo_export_optimizer(aleppo_ctx *c, float *exp_avg, float *exp_avg_sq, int64_t *step,
                                       size_t count) {
  CHECK_CTX(c);
  if (!step)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null step");
  int rc = export_flat(c, c->M1, exp_avg, count);
  if (rc)
    return rc;
  rc = export_flat(c, c->M2, exp_avg_sq, count);
  if (rc)
    return rc;
  *step = c->adam_step;
  return ALEPPO_OK;
}
extern "C" int aleppo_import_optimizer(aleppo_ctx *c, const float *exp_avg, const float *exp_avg_sq, int64_t step,
                                       size_t count) {
  CHECK_CTX(c);
  if (!exp_avg || !exp_avg_sq || count != c->L.reference_count() || step < 0)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "import_optimizer: bad argument");
  std::vector<float> tmp(c->L.total());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  params_to_internal(c->L, exp_avg, tmp.data());
  HIPCHK(c, copy_sync(c, c->M1, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
  params_to_internal(c->L, exp_avg_sq, tmp.data());
  HIPCHK(c, copy_sync(c, c->M2, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
  c->adam_step = step;
  return ALEPPO_OK;
}

// ------------------------------------------------------------------ rollout
static int do_act(aleppo_ctx *c, const float *noise, int slot, void *logits_dst, void *values_dst, int *actions_dst,
                  bool publish) {
  if (c->pre_acted != slot) { // conv stack + split-K fc at acting size; the head kernel finishes the fc reduction
    const SampleMap map = slot_map(c, slot);
    if (c->prec == ALEPPO_BF16 && use_patch_kernels()) { // one launch: a1/a2 never leave LDS
      prof_begin(c, ALEPPO_K_CONV1_FWD);
      patch_act_convs(c->stream, c->obs, map, Pcw(c, P_W1), Pf(c, P_B1), Pcw(c, P_W2), Pf(c, P_B2), Pcw(c, P_W3),
                      Pf(c, P_B3), c->a3, c->E);
      prof_end(c, ALEPPO_K_CONV1_FWD);
    } else {
    prof_begin(c, ALEPPO_K_CONV1_FWD);
    conv1_fwd(c->stream, c->prec, c->obs, map, Pcw(c, P_W1), Pf(c, P_B1), c->a1, c->E);
    prof_end(c, ALEPPO_K_CONV1_FWD);
    prof_begin(c, ALEPPO_K_CONV2_FWD);
    conv2_fwd(c->stream, c->prec, c->a1, Pcw(c, P_W2), Pf(c, P_B2), c->a2, c->E);
    prof_end(c, ALEPPO_K_CONV2_FWD);
    prof_begin(c, ALEPPO_K_CONV3_FWD);
    conv3_fwd(c->stream, c->prec, c->a2, Pcw(c, P_W3), Pf(c, P_B3), c->a3, c->E);
    prof_end(c, ALEPPO_K_CONV3_FWD);
    }
    prof_begin(c, ALEPPO_K_FC_FWD);
    fc_fwd_splitk(c->stream, c->prec, c->a3, Pcw(c, P_WFC), c->hpart, c->E, c->H);
    prof_end(c, ALEPPO_K_FC_FWD);
  }
  c->pre_acted = -1; // (consumed; a3 / hpart are scratch again)
  const float *dn = nullptr;
  if (noise) { // (two staging halves: with a gated replay the next slot is enqueued before this copy has run)
    const size_t half = (size_t)(c->noise_flip++ & 1u) * c->E * c->A;
    std::memcpy(c->h_noise + half, noise, (size_t)c->E * c->A * 4);
    HIPCHK(c, hipMemcpyAsync(c->d_noise + half, c->h_noise + half, (size_t)c->E * c->A * 4, hipMemcpyHostToDevice,
                             c->stream));
    dn = c->d_noise + half;
  }
  int64_t *pinned_dev = nullptr;
  HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&pinned_dev), c->h_actions, 0));
  prof_begin(c, ALEPPO_K_INFER_HEAD);
  if (publish)
    c->ticket++;
  if (c->dbg_no_publish)
    pinned_dev = nullptr;
  launch_infer_head(c->stream, c->hpart, FC_SPLITS, Pf(c, P_BFC), Pf(c, P_WH), Pf(c, P_BH), dn, c->cfg.seed,
                    c->rng_counter++, logits_dst, values_dst, actions_dst, pinned_dev, publish ? c->d_done : nullptr,
                    c->ticket, c->E, c->H, c->A, nullptr, c->rt16);
  prof_end(c, ALEPPO_K_INFER_HEAD);
  HIPCHK(c, hipGetLastError());
  return ALEPPO_OK;
}

// enqueue slot c->t's acting kernels (whatever aleppo_step has not already run) + the head that publishes the actions
static int act_enqueue(aleppo_ctx *c, const float *noise, int slot) {
  if (slot >= c->T)
    return set_err(c, ALEPPO_ERR_RUNTIME, "rollout buffer is full: call aleppo_finish_rollout");
  if (slot == 0 && c->need_carry) { // slot T of the previous rollout is this rollout's first observation
    launch_copy_slot(c->stream, c->obs, c->E, c->T + 1, c->T, 0);
    c->need_carry = false;
  }
  const size_t o = (size_t)slot * c->E;
  return do_act(c, noise, slot, rp(c, c->logits_tm, o * c->A), rp(c, c->values_tm, o), c->actions_tm + o, true);
}
// The gate's exit condition fired: the stream ran (or will run) a slot whose frames the host never released.
static int check_gate(aleppo_ctx *c) {
  const unsigned long long rep = __atomic_load_n(c->h_go + 1, __ATOMIC_ACQUIRE);
  if (!rep)
    return ALEPPO_OK;
  char b[256];
  std::snprintf(b, sizeof b,
                "the slot-ahead gate %llu was not released within %.0f ms (released so far: %llu): the device went on "
                "without the host's frames",
                rep, (double)c->gate_timeout_ticks / 1e5, __atomic_load_n(c->h_go, __ATOMIC_RELAXED));
  return fail_ctx(c, ALEPPO_ERR_RUNTIME, b);
}
// wait for the ticket the head kernel publishes after the actions (bounded spin, then a real sync)
// stream_parked: the stream already holds the NEXT slot behind the release word (gated replay) - a stream sync would
// wait for a release only this thread can give, so the wait only spins (and yields once the slot is clearly a long one)
static int act_wait(aleppo_ctx *c, long long ticket, bool stream_parked = false) {
  volatile long long *tk = reinterpret_cast<volatile long long *>(c->h_actions + c->E);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  bool slow = false;
  if (c->dbg_no_publish)
    HIPCHK(c, hipStreamSynchronize(c->stream));
  while (!c->dbg_no_publish && *tk != ticket) {
    if (slow)
      std::this_thread::yield();
    else
      __builtin_ia32_pause();
    if ((++spins & 1023u) == 0) {
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (slow) {
        if (int rc = check_gate(c))
          return rc;
        // Backstop only (no hand-off of this design waits on anything but the head kernel in front of it): say what the
        // stream and the hand-off words look like, so that a missing ticket can be told from a stuck stream.
        if (waited > 60.0 + (double)c->gate_timeout_ticks / 1e8) {
          const hipError_t q = hipStreamQuery(c->stream);
          unsigned int done = 0xFFFFFFFFu; // the head's arrival counter, read on the (idle) side stream
          if ((c->comm_stream || hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) == hipSuccess) &&
              hipMemcpyAsync(c->h_err + 1, c->d_done, 4, hipMemcpyDeviceToHost, c->comm_stream) == hipSuccess) {
            for (int i = 0; i < 1000 && hipStreamQuery(c->comm_stream) == hipErrorNotReady; ++i)
              std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (hipStreamQuery(c->comm_stream) == hipSuccess)
              done = (unsigned int)c->h_err[1];
          }
          char b[384];
          std::snprintf(b, sizeof b,
                        "the acting head's ticket did not arrive within %.0f s: expected %lld, pinned word %lld, release "
                        "word %llu, last gate %llu, gate report %llu, head arrival counter %u of %d, stream %s",
                        waited, ticket, (long long)*tk, __atomic_load_n(c->h_go, __ATOMIC_RELAXED), c->go_seq,
                        __atomic_load_n(c->h_go + 1, __ATOMIC_RELAXED), done, (c->E + 3) / 4,
                        q == hipSuccess ? "idle" : q == hipErrorNotReady ? "busy" : hipGetErrorString(q));
          return fail_ctx(c, ALEPPO_ERR_RUNTIME, b);
        }
      } else if (waited > 2e-3) {
        if (stream_parked) {
          slow = true;
          continue;
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        break;
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return check_gate(c);
}
extern "C" int aleppo_act(aleppo_ctx *c, const float *noise, const int64_t **actions_pinned) {
  CHECK_CTX(c);
  int rc = ALEPPO_OK;
  if (c->act_queued_slot == c->t && c->t < c->T) { // enqueued by aleppo_arm_step (with ITS noise): only the wait is left
    c->act_queued_slot = -1;
    rc = act_wait(c, c->act_queued_ticket);
  } else {
    rc = act_enqueue(c, noise, c->t);
    if (rc == ALEPPO_OK)
      rc = act_wait(c, c->ticket);
  }
  if (rc)
    return rc;
  if (actions_pinned)
    *actions_pinned = c->h_actions;
  return ALEPPO_OK;
}

static int upload_frames(aleppo_ctx *c, const uint8_t *frames, int kind, int location, const uint8_t **dev_frames) {
  const size_t bytes = (size_t)c->E * (kind == ALEPPO_FRAMES_RAW_PAIR ? 2 * RAW_H * RAW_W : FRAME_PIX);
  if (location == ALEPPO_DEVICE) {
    if (reinterpret_cast<uintptr_t>(frames) % 16)
      return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "device frames must be 16-byte aligned");
    *dev_frames = frames;
    return ALEPPO_OK;
  }
  if (location == ALEPPO_HOST_MAPPED) { // the kernel reads the page-locked host buffer in place
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<uint8_t *>(frames), 0) != hipSuccess || !dp)
      return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT,
                     "ALEPPO_HOST_MAPPED frames must lie in mapped page-locked host memory (hipHostMalloc / hipHostRegister)");
    if (reinterpret_cast<uintptr_t>(dp) % 16)
      return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "mapped frames must be 16-byte aligned");
    *dev_frames = static_cast<const uint8_t *>(dp);
    return ALEPPO_OK;
  }
  if (location != ALEPPO_HOST)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "unknown frame location");
  std::memcpy(c->h_frames, frames, bytes);
  HIPCHK(c, hipMemcpyAsync(c->d_frames, c->h_frames, bytes, hipMemcpyHostToDevice, c->stream));
  *dev_frames = c->d_frames;
  return ALEPPO_OK;
}

static int do_push(aleppo_ctx *c, const uint8_t *frames, int kind, int location, const uint8_t *episode_start) {
  if (!frames || !episode_start)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  if (kind != ALEPPO_FRAMES_84 && kind != ALEPPO_FRAMES_RAW_PAIR)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "unknown frame kind");
  if (c->t >= c->T)
    return set_err(c, ALEPPO_ERR_RUNTIME, "rollout buffer is full: call aleppo_finish_rollout");
  // staging buffers are reused every step: wait for the previous upload (normally long finished: act() syncs)
  HIPCHK(c, hipEventSynchronize(c->ev_tmp));
  uint8_t *hs = c->h_step + c->step_rec_bytes;
  std::memcpy(hs, episode_start, c->E);
  HIPCHK(c, hipMemcpyAsync(c->d_start, hs, c->E, hipMemcpyHostToDevice, c->stream));
  const uint8_t *df = nullptr;
  int rc = upload_frames(c, frames, kind, location, &df);
  if (rc)
    return rc;
  c->pre_acted = -1;
  prof_begin(c, ALEPPO_K_INGEST);
  launch_ingest(c->stream, kind == ALEPPO_FRAMES_RAW_PAIR, df, c->lut, c->d_start, nullptr, c->obs, c->E, c->T + 1,
                c->t, c->t + 1);
  prof_end(c, ALEPPO_K_INGEST);
  HIPCHK(c, hipGetLastError());
  return ALEPPO_OK;
}
static int do_record(aleppo_ctx *c, const float *rewards, const uint8_t *terminated, const uint8_t *truncated,
                     const uint8_t *episode_start) {
  if (!rewards || !terminated || !truncated || !episode_start)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  if (c->t >= c->T)
    return set_err(c, ALEPPO_ERR_RUNTIME, "rollout buffer is full: call aleppo_finish_rollout");
  const int E = c->E;
  std::memcpy(c->h_step, rewards, (size_t)E * 4);
  std::memcpy(c->h_step + 4 * (size_t)E, terminated, E);
  std::memcpy(c->h_step + 5 * (size_t)E, truncated, E);
  std::memcpy(c->h_step + 6 * (size_t)E, episode_start, E);
  std::memcpy(c->h_rec + (size_t)c->t * c->step_rec_bytes, c->h_step, (size_t)7 * E); // uploaded at finish_rollout
  c->t++;
  return ALEPPO_OK;
}

// the GPU side of a step: slot t's new frames -> observation slot t + 1 (+ its convolutions and fc where fused).  The
// episode-start flags come as a kernel-argument bitmask (sb) or, for a step enqueued before the emulator has produced
// them (aleppo_arm_step), as bytes in mapped host memory (start_mapped: host pointer, start_dev: its device address).
static int step_enqueue(aleppo_ctx *c, const uint8_t *df, int kind, int location, const StartBits *sb,
                        const uint8_t *start_mapped, int t) {
  const int E = c->E;
  uint8_t *start_dev = nullptr;
  if (start_mapped)
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&start_dev), const_cast<uint8_t *>(start_mapped), 0));
  // Fused ingest pays for given 84x84 frames (15.7 -> 14.4 ms per 128-slot rollout+update with the frames in mapped host
  // memory) and for raw pairs read over the bus (one environment's pair is staged by ONE workgroup with nine 16-byte
  // loads per thread in flight).  Raw pairs resident in HBM stay on the stand-alone ingest kernel: there the palette
  // lookups of 67 K source bytes per environment are spread over all 256 CUs instead of the 128 acting workgroups
  // (measured: 5.01 vs 5.10 ms per rollout).
  const bool fuse = c->prec == ALEPPO_BF16 && use_patch_kernels() && c->tune.fused_act &&
                    !(kind == ALEPPO_FRAMES_RAW_PAIR && location != ALEPPO_HOST_MAPPED && c->tune.fused_act < 2);
  if (fuse) {
    // ONE launch forms slot t+1's stack from the new frames AND runs conv1 -> conv2 -> conv3 on it, then the split-K fc:
    // when the next aleppo_act (or aleppo_finish_rollout's bootstrap) arrives only the head + sampling kernel is left.
    // A slot's critical path is 3 dependent launches instead of 4 and the stack skips one HBM round trip.
    const SampleMap map = slot_map(c, t + 1);
    prof_begin(c, ALEPPO_K_ACT_FUSED);
    patch_act_convs(c->stream, c->obs, map, Pcw(c, P_W1), Pf(c, P_B1), Pcw(c, P_W2), Pf(c, P_B2), Pcw(c, P_W3),
                    Pf(c, P_B3), c->a3, E, kind == ALEPPO_FRAMES_RAW_PAIR ? 2 : 1, df, c->lut, sb, -(long)FRAME_PIX,
                    start_dev);
    prof_end(c, ALEPPO_K_ACT_FUSED);
    prof_begin(c, ALEPPO_K_FC_FWD);
    fc_fwd_splitk(c->stream, c->prec, c->a3, Pcw(c, P_WFC), c->hpart, E, c->H);
    prof_end(c, ALEPPO_K_FC_FWD);
    c->pre_acted = t + 1;
  } else {
    if (start_mapped) // (every thread of the stand-alone kernel reads its flag: from HBM, not across the bus)
      HIPCHK(c, hipMemcpyAsync(c->d_start, start_mapped, E, hipMemcpyHostToDevice, c->stream));
    prof_begin(c, ALEPPO_K_INGEST);
    launch_ingest(c->stream, kind == ALEPPO_FRAMES_RAW_PAIR, df, c->lut, start_mapped ? c->d_start : nullptr, sb, c->obs, E,
                  c->T + 1, t, t + 1);
    prof_end(c, ALEPPO_K_INGEST);
    c->pre_acted = -1;
  }
  return ALEPPO_OK;
}

extern "C" int aleppo_push_frames(aleppo_ctx *c, const uint8_t *frames, int kind, int location,
                                  const uint8_t *episode_start) {
  CHECK_CTX(c);
  int rc = do_push(c, frames, kind, location, episode_start);
  if (rc == ALEPPO_OK)
    HIPCHK(c, hipEventRecord(c->ev_tmp, c->stream));
  return rc;
}
extern "C" int aleppo_record_step(aleppo_ctx *c, const float *rewards, const uint8_t *terminated,
                                  const uint8_t *truncated, const uint8_t *episode_start) {
  CHECK_CTX(c);
  HIPCHK(c, hipEventSynchronize(c->ev_tmp));
  return do_record(c, rewards, terminated, truncated, episode_start);
}
extern "C" int aleppo_step(aleppo_ctx *c, const uint8_t *frames, int kind, int location, const float *rewards,
                           const uint8_t *terminated, const uint8_t *truncated, const uint8_t *episode_start) {
  CHECK_CTX(c);
  if (!frames || !rewards || !terminated || !truncated || !episode_start)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  if (kind != ALEPPO_FRAMES_84 && kind != ALEPPO_FRAMES_RAW_PAIR)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "unknown frame kind");
  if (c->t >= c->T)
    return set_err(c, ALEPPO_ERR_RUNTIME, "rollout buffer is full: call aleppo_finish_rollout");
  // No per-slot upload: the scalars of rollout.cc:212-227 are packed into a pinned host record and reach the
  // device in ONE copy at finish_rollout (only GAE reads them); the episode-start flags ingest needs now
  // travel as a kernel-argument bitmask.
  const int E = c->E;
  uint8_t *rec = c->h_rec + (size_t)c->t * c->step_rec_bytes;
  std::memcpy(rec, rewards, (size_t)E * 4);
  std::memcpy(rec + 4 * (size_t)E, terminated, E);
  std::memcpy(rec + 5 * (size_t)E, truncated, E);
  std::memcpy(rec + 6 * (size_t)E, episode_start, E);
  StartBits sb{};
  for (int e = 0; e < E; ++e)
    if (episode_start[e])
      sb.w[e >> 5] |= 1u << (e & 31);
  const uint8_t *df = nullptr;
  if (location == ALEPPO_HOST)
    HIPCHK(c, hipEventSynchronize(c->ev_tmp)); // frame staging reuse guard
  int rc = upload_frames(c, frames, kind, location, &df);
  if (rc)
    return rc;
  rc = step_enqueue(c, df, kind, location, &sb, nullptr, c->t);
  if (rc)
    return rc;
  HIPCHK(c, hipGetLastError());
  if (location == ALEPPO_HOST)
    HIPCHK(c, hipEventRecord(c->ev_tmp, c->stream));
  c->t++;
  return ALEPPO_OK;
}
// Park the main stream: everything enqueued after this runs once the host has stored a number >= the returned sequence
// number into the release word (gate_kernel; it gives up after gate_timeout_ticks and reports, see check_gate).
static int gate_enqueue(aleppo_ctx *c) {
  unsigned long long *go_dev = nullptr;
  HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&go_dev), c->h_go, 0));
  launch_gate(c->stream, go_dev, c->go_seq + 1, c->gate_timeout_ticks);
  HIPCHK(c, hipGetLastError());
  c->go_seq++; // (only once the gate is on the stream: an error above leaves nothing to release)
  return ALEPPO_OK;
}
static inline void gate_release(aleppo_ctx *c) { __atomic_store_n(c->h_go, c->go_seq, __ATOMIC_RELEASE); }

// Live loops one slot ahead (the replay loop below does the same with a recorded trace): see include/aleppo.h.
extern "C" int aleppo_arm_step(aleppo_ctx *c, const uint8_t *frames, int kind, const uint8_t *episode_start_mapped,
                               const float *noise_next) {
  CHECK_CTX(c);
  if (!frames || !episode_start_mapped)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  if (kind != ALEPPO_FRAMES_84 && kind != ALEPPO_FRAMES_RAW_PAIR)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "unknown frame kind");
  if (c->t >= c->T)
    return set_err(c, ALEPPO_ERR_RUNTIME, "rollout buffer is full: call aleppo_finish_rollout");
  if (c->prof_on || c->dbg_no_publish)
    return set_err(c, ALEPPO_ERR_RUNTIME, "aleppo_arm_step is not available while per-kernel profiling is on");
  const uint8_t *df = nullptr;
  int rc = upload_frames(c, frames, kind, ALEPPO_HOST_MAPPED, &df);
  if (rc)
    return rc;
  void *sdev = nullptr;
  if (hipHostGetDevicePointer(&sdev, const_cast<uint8_t *>(episode_start_mapped), 0) != hipSuccess || !sdev)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT,
                   "episode_start_mapped must lie in mapped page-locked host memory (aleppo_host_alloc)");
  rc = gate_enqueue(c);
  if (rc)
    return rc;
  rc = step_enqueue(c, df, kind, ALEPPO_HOST_MAPPED, nullptr, episode_start_mapped, c->t);
  if (rc == ALEPPO_OK && c->t + 1 < c->T) {
    rc = act_enqueue(c, noise_next, c->t + 1);
    c->act_queued_slot = c->t + 1;
    c->act_queued_ticket = c->ticket;
  }
  if (rc || hipGetLastError() != hipSuccess) {
    // part of the slot is on the stream behind the gate and must not run on frames that do not exist yet: the context
    // is failed (sticky), the gate released so that the stream drains (the caller's buffers stay referenced until
    // aleppo_destroy)
    const std::string why = rc ? c->err : std::string("a launch failed");
    return fail_ctx(c, rc ? rc : ALEPPO_ERR_HIP, "aleppo_arm_step: " + why);
  }
  c->armed = true;
  c->armed_start = episode_start_mapped;
  return ALEPPO_OK;
}
extern "C" int aleppo_release_step(aleppo_ctx *c, const float *rewards, const uint8_t *terminated,
                                   const uint8_t *truncated) {
  CHECK_CTX_ANY(c);
  if (!c->armed)
    return set_err(c, ALEPPO_ERR_RUNTIME, "aleppo_release_step without an armed step");
  if (!rewards || !terminated || !truncated) // (checked while still armed: the caller can repeat the call)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  if (int rc = check_gate(c)) // the gate gave up before this release: the slot already ran without the frames
    return rc;
  // the frames and the episode-start bytes are in place: let the stream go FIRST, the bookkeeping is off its path
  std::atomic_thread_fence(std::memory_order_release);
  gate_release(c);
  c->armed = false;
  const int E = c->E;
  uint8_t *rec = c->h_rec + (size_t)c->t * c->step_rec_bytes; // uploaded at finish_rollout
  std::memcpy(rec, rewards, (size_t)E * 4);
  std::memcpy(rec + 4 * (size_t)E, terminated, E);
  std::memcpy(rec + 5 * (size_t)E, truncated, E);
  std::memcpy(rec + 6 * (size_t)E, c->armed_start, E);
  c->t++;
  return ALEPPO_OK;
}
extern "C" int aleppo_replay_rollout(aleppo_ctx *c, const uint8_t *frames, int kind, int location,
                                     size_t slot_stride_bytes, const float *rewards, const uint8_t *terminated,
                                     const uint8_t *truncated, const uint8_t *episode_start, const float *noise) {
  CHECK_CTX(c);
  if (!frames || !rewards || !terminated || !truncated || !episode_start)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  if (location != ALEPPO_DEVICE && location != ALEPPO_HOST_MAPPED)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "replay_rollout: frames must be ALEPPO_DEVICE or ALEPPO_HOST_MAPPED");
  if (c->t != 0)
    return set_err(c, ALEPPO_ERR_RUNTIME, "replay_rollout needs an empty rollout buffer");
  const size_t E = (size_t)c->E;
  // rollout.cc:198-278 with the emulator replaced by the recorded trace.  The stream runs ONE slot ahead of the host:
  // slot t + 1's kernels (ingest of the frames the emulator produces from action t, convolutions, fc, head) are enqueued
  // while slot t is still on the GPU, behind a gate (gate_kernel) on a pinned word that the host writes once it HAS
  // slot t's actions (and, with a live emulator, the frames).  The hand-off keeps its order - the GPU never touches slot
  // t + 1's frames before the host has seen action t - but the next slot starts ~1 us after the host's store instead of
  // a kernel-launch latency after it (micro-benchmark tests/tools/waitvalue.hip: 3.7 vs 6.7 us per ping-pong).
  static const bool gated_env = [] {
    const char *e = getenv("ALEPPO_REPLAY_GATED");
    return !e || atoi(e) != 0;
  }();
  const bool gated = gated_env && !c->prof_on && !c->dbg_no_publish;
  auto noise_at = [&](int t) { return noise ? noise + (size_t)t * E * c->A : nullptr; };
  int rc = act_enqueue(c, noise_at(0), 0);
  if (rc)
    return rc;
  for (int t = 0; t < c->T; ++t) {
    const long long ticket_t = c->ticket; // of act(t), enqueued above / in the previous iteration
    if (gated) {
      rc = gate_enqueue(c); // (nothing of this slot is behind a gate yet: an ordinary error)
      if (rc)
        return rc;
    } else {
      rc = act_wait(c, ticket_t);
      if (rc)
        return rc;
    }
    rc = aleppo_step(c, frames + (size_t)t * slot_stride_bytes, kind, location, rewards + (size_t)t * E,
                     terminated + (size_t)t * E, truncated + (size_t)t * E, episode_start + (size_t)t * E);
    if (rc == ALEPPO_OK && t + 1 < c->T)
      rc = act_enqueue(c, noise_at(t + 1), t + 1);
    if (gated) {
      if (rc) // a slot is half enqueued behind the gate: see aleppo_arm_step
        return fail_ctx(c, rc, "aleppo_replay_rollout: " + c->err);
      rc = act_wait(c, ticket_t, /*stream_parked=*/true); // the host has slot t's actions: the emulator would step now
      if (rc)
        return c->failed ? rc : fail_ctx(c, rc, "aleppo_replay_rollout: " + c->err);
      gate_release(c);            // ... and hand over slot t + 1's frames
    } else if (rc) {
      return rc;
    }
  }
  return ALEPPO_OK;
}
extern "C" int aleppo_host_alloc(aleppo_ctx *c, size_t bytes, void **ptr) {
  CHECK_CTX(c);
  if (!ptr || bytes == 0)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "host_alloc: bad argument");
  *ptr = nullptr;
  HIPCHK(c, hipHostMalloc(ptr, bytes, hipHostMallocMapped));
  std::memset(*ptr, 0, bytes);
  return ALEPPO_OK;
}
extern "C" int aleppo_host_free(aleppo_ctx *c, void *ptr) {
  CHECK_CTX(c);
  if (!ptr)
    return ALEPPO_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream)); // a kernel may still be reading it
  HIPCHK(c, hipHostFree(ptr));
  return ALEPPO_OK;
}
extern "C" int aleppo_set_gray_lut(aleppo_ctx *c, const uint8_t *lut256) {
  CHECK_CTX(c);
  if (!lut256)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null lut");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_sync(c, c->lut, lut256, 256, hipMemcpyHostToDevice));
  return ALEPPO_OK;
}

extern "C" int aleppo_finish_rollout(aleppo_ctx *c, const float *noise) {
  CHECK_CTX(c);
  c->act_queued_slot = -1;
  if (c->t != c->T)
    return set_err(c, ALEPPO_ERR_RUNTIME, "Buffer is not full, cannot compute GAE."); // buffer.cc:64-65
  if (int rcg = check_gate(c))
    return rcg;
  const int E = c->E, T = c->T, A = c->A;
  // extra selector call on the post-rollout observation: its values bootstrap slot T-1, its sample is
  // discarded but advances the RNG stream like the reference (rollout.cc:268-270)
  int rc = do_act(c, noise, T, rp(c, c->logits_tm, (size_t)T * E * A), rp(c, c->values_tm, (size_t)T * E),
                  c->actions_tm + (size_t)T * E, false);
  if (rc)
    return rc;
  HIPCHK(c, hipMemcpyAsync(c->step_rec, c->h_rec, c->step_rec_bytes * T, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_err, 0, 4, c->stream));
  prof_begin(c, ALEPPO_K_GAE);
  launch_gae(c->stream, c->step_rec, c->step_rec_bytes, c->values_tm, c->logits_tm, c->actions_tm, c->adv_n, c->ret_n,
             c->oldlp_n, c->act_n, c->mask_n, c->d_err, E, T, A, c->cfg.gamma, c->cfg.lambda, true, c->rt16);
  prof_end(c, ALEPPO_K_GAE);
  if (c->cfg.advantage_norm) {
    launch_adv_norm(c->stream, c->adv_n, c->mask_n, c->adv_stats, c->N, 0, c->rt16);
    if ((c->world > 1 || c->force_comm) && c->nccl_comm)
      NCCLCHK(c, ncclAllReduce(c->adv_stats, c->adv_stats, 3, ncclFloat, ncclSum,
                               static_cast<ncclComm_t>(c->nccl_comm), c->stream));
    launch_adv_norm(c->stream, c->adv_n, c->mask_n, c->adv_stats, c->N, 1, c->rt16);
  }
  HIPCHK(c, hipMemcpyAsync(c->h_err, c->d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (int rcg = check_gate(c)) // (the last armed slot's gate may have given up while the stream drained)
    return rcg;
  c->t = 0;
  c->pre_acted = -1;
  c->need_carry = true;
  c->batch_n = c->N;
  if (*c->h_err)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT,
                   "Episode starts, terminals, and truncations must be mutually exclusive."); // gae.cc:49-53
  return ALEPPO_OK;
}

// ------------------------------------------------------------------ update
static int ensure_metric_storage(aleppo_ctx *c, int epochs, int M, long B) {
  const size_t need = (size_t)epochs * M * B;
  if (need > c->metric_cap) {
    retire(c, c->metric_ps); // (never hipFree before aleppo_destroy: see dalloc)
    c->metric_ps = nullptr;
    HIPCHK(c, dalloc(&c->metric_ps, need * 5 * 4, c->stream));
    c->metric_cap = need;
  }
  const size_t nm = (size_t)epochs * M;
  if (nm > c->metric_red_cap) {
    retire(c, c->metric_red);
    retire(c, c->grad_norms);
    retire(c, c->adam_sched);
    retire_host(c, c->h_metric_red);
    retire_host(c, c->h_adam_sched);
    c->metric_red = c->grad_norms = c->h_metric_red = c->adam_sched = c->h_adam_sched = nullptr;
    HIPCHK(c, dalloc(&c->metric_red, nm * 8 * 4, c->stream));
    HIPCHK(c, dalloc(&c->grad_norms, nm * 4, c->stream));
    HIPCHK(c, dalloc(&c->adam_sched, nm * 2 * 4, c->stream));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_metric_red), nm * 9 * 4, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_adam_sched), nm * 2 * 4, hipHostMallocDefault));
    c->metric_red_cap = nm;
  }
  return ALEPPO_OK;
}

static bool fuse_tail_env() { // A/B switch: conv1's slab reduce fused into the sum-of-squares pass (default)
  static const bool v = [] {
    const char *e = getenv("ALEPPO_FUSE_TAIL_REDUCE");
    return !e || atoi(e) != 0;
  }();
  return v;
}

extern "C" int aleppo_train(aleppo_ctx *c, double lr, int epochs, int M, aleppo_minibatch_metrics *out) {
  CHECK_CTX(c);
  if (epochs <= 0 || M <= 0)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "epochs and num_mini_batches must be positive");
  const long N = c->batch_n;
  if (N <= 0)
    return set_err(c, ALEPPO_ERR_RUNTIME, "no batch: call aleppo_finish_rollout or aleppo_set_batch first");
  c->pre_acted = -1; // the update's activations overwrite the acting scratch, and the weights change
  if (N % M != 0)
    return set_err(c, ALEPPO_ERR_RUNTIME, "Batch size must be divisible by num_mini_batches"); // train.h:140-143
  const long B = N / M;
  if (B > c->maxB)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "minibatch larger than config.max_minibatch");
  if (M > 4096)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "num_mini_batches > 4096");
  if (c->world > 1 && !c->nccl_comm)
    return set_err(c, ALEPPO_ERR_RUNTIME, "world_size > 1 but aleppo_comm_init was not called");
  int rc = ensure_metric_storage(c, epochs, M, B);
  if (rc)
    return rc;
  ncclComm_t comm = static_cast<ncclComm_t>(c->nccl_comm);
  const bool dp = c->world > 1 || (c->nccl_comm && c->force_comm); // force_comm: 1-rank communicator (tests)
  // side streams, created on first use (see aleppo_create): the weight-gradient stream, and - only with data parallelism -
  // the communication stream
  if (!c->wg_stream)
    HIPCHK(c, hipStreamCreateWithFlags(&c->wg_stream, hipStreamNonBlocking));
  if (dp && !c->comm_stream)
    HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  const int H = c->H, A = c->A, prec = c->prec;
  const ParamLayout &L = c->L;
  hipStream_t s = c->stream;
  const Hyper hp{c->cfg.clip_param, c->cfg.value_loss_coef, c->cfg.entropy_coef, c->cfg.max_gradient_norm};

  // Adam's per-step scalars for the whole call, uploaded once: step size lr / (1 - beta1^t) and sqrt(1 - beta2^t) are
  // DEVICE values the Adam kernel reads (kernel arguments would be baked into a captured graph)
  const int nm = epochs * M;
  for (int i = 0; i < nm; ++i) {
    const double b1 = c->cfg.adam_beta1, b2 = c->cfg.adam_beta2, t = (double)(c->adam_step + i + 1);
    c->h_adam_sched[2 * i] = (float)(lr / (1.0 - std::pow(b1, t)));
    c->h_adam_sched[2 * i + 1] = (float)std::sqrt(1.0 - std::pow(b2, t));
  }
  HIPCHK(c, hipMemcpyAsync(c->adam_sched, c->h_adam_sched, (size_t)nm * 8, hipMemcpyHostToDevice, s));

  float *sW1 = c->slab + c->slab_off[0], *sB1 = c->slab + c->slab_off[1], *sW2 = c->slab + c->slab_off[2],
        *sB2 = c->slab + c->slab_off[3], *sW3 = c->slab + c->slab_off[4], *sB3 = c->slab + c->slab_off[5],
        *sWfc = c->slab + c->slab_off[6], *sBfc = c->slab + c->slab_off[7], *sWh = c->slab + c->slab_off[8],
        *sBh = c->slab + c->slab_off[9];
  const size_t fs = c->metric_cap; // field stride of the per-sample metric arrays
  const int nblk_head = (int)std::min<long>(MAXS_HEAD, (B + 15) / 16);
  const int nblk_sq = (int)std::min<size_t>(800, (L.off[P_W1] + 4095) / 4096); // + 129 conv1 chunks <= 1024 partials

  // The three weight-gradient kernels (and the slab reduce of bucket 0) run on their own stream
  // next to the dgrad chain (fc wgrad || fc dgrad, conv3 wgrad || conv3 dgrad, conv2 wgrad || conv2 dgrad -> conv1
  // wgrad).  Every kernel is a latency-bound full-GPU persistent grid: co-scheduling fills the drain / ramp bubbles
  // between dependent launches.
  static const bool two_env = [] { // process-wide A/B switch, read once
    const char *e = std::getenv("ALEPPO_BWD_STREAMS");
    return !(e && std::atoi(e) == 1); // ALEPPO_BWD_STREAMS=1: everything on one stream (A/B testing: 8.80 ms)
  }();
  const bool two = two_env && !c->serial_update; // (profiling brackets every kernel on the stream it runs on)
  // (Not with data parallelism: the fused kernel's 512-register workgroups need whole CUs, and bucket 0's all-reduce - whose
  // RCCL kernels are resident on some of them by the time the dgrad chain gets there - is what the conv backward is meant to
  // run BESIDE; the three launches share CUs with it, the fused kernel's workgroups on those CUs would start when it ends.)
  const bool bwd_fused = (c->tune.fused_bwd == 2 || (c->tune.fused_bwd == 1 && B >= 2048)) && prec == ALEPPO_BF16 &&
                         use_patch_kernels() && !dp;
  hipStream_t sw = two ? c->wg_stream : s; // stream of the weight-gradient kernels
  // (Tried in round 3, tests/tools/forkbench.hip: in isolation an event record + wait costs the pair of streams ~12 us
  // per dependency, a one-wave signal kernel + a one-wave gate kernel on a device word ~3 us.  In the update it changes
  // nothing or loses: with signal kernels the weight-gradient kernel is released BEFORE the dgrad kernel beside it has its
  // workgroups on the CUs and the dgrad chain slows down (485 vs 458 us per minibatch); with the dgrad kernel itself
  // announcing its start the main stream runs without gaps - and the minibatch takes 445 vs 444 us: the two streams
  // together keep the GPU saturated, so a gap on one is filled by the other.  DESIGN.md 4a.)
  auto fork = [&](hipEvent_t ev) -> hipError_t { // sw continues after everything enqueued on s so far
    if (!two)
      return hipSuccess;
    const hipError_t e = hipEventRecord(ev, s);
    return e != hipSuccess ? e : hipStreamWaitEvent(sw, ev, 0);
  };
  // Everything the update enqueues - mask counts, epochs x minibatches of forward / loss / backward / [all-reduce] /
  // clip / Adam, the metric reduction - as one function: run eagerly, or recorded once into a hipGraph and replayed.
  auto enqueue_update = [&]() -> int {
  launch_mask_count(s, c->mask_n, c->mask_counts, B, M);
  if (dp) // N_m of the masked mean is the GLOBAL count
    NCCLCHK(c, ncclAllReduce(c->mask_counts, c->mask_counts, M, ncclFloat, ncclSum, comm, s));
  for (int ep = 0; ep < epochs; ++ep)
    for (int mb = 0; mb < M; ++mb) { // contiguous env-major slices; randperm unused (Q1)
      const int mi = ep * M + mb;
      const long n0 = (long)mb * B;
      const SampleMap map = train_map(c, n0);
      const int hparts = net_forward(c, c->obs, map, B, FC_FWD_MAX_PARTS);
      prof_begin(c, ALEPPO_K_HEAD);
      launch_head_train(s, c->h, Pf(c, P_WH), Pf(c, P_BH), c->act_n + n0, rp(c, c->oldlp_n, (size_t)n0 * A),
                        rp(c, c->adv_n, (size_t)n0), rp(c, c->ret_n, (size_t)n0), c->mask_n + n0, c->mask_counts + mb, hp,
                        c->dh, prec,
                        c->metric_ps + 0 * fs + (size_t)mi * B, c->metric_ps + 1 * fs + (size_t)mi * B,
                        c->metric_ps + 2 * fs + (size_t)mi * B, c->metric_ps + 3 * fs + (size_t)mi * B,
                        c->metric_ps + 4 * fs + (size_t)mi * B, sWh, sBh, nblk_head, B, H, A, nullptr, nullptr, hparts,
                        c->rt16);
      prof_end(c, ALEPPO_K_HEAD);
      HIPCHK(c, fork(c->ev_head)); // dh is ready
      prof_begin(c, ALEPPO_K_FC_DGRAD);
      fc_dgrad(s, prec, c->dh, c->WfcT, c->a3, c->dz3, B, H);
      prof_end(c, ALEPPO_K_FC_DGRAD);
      prof_begin(c, ALEPPO_K_FC_WGRAD, sw);
      // split-K slabs (or, with one slice, straight into the gradient tensor)
      const bool fc_direct = fc_wgrad_slices(prec, B) == 1;
      const int Sfc = fc_wgrad(sw, prec, c->dh, c->a3, fc_direct ? c->G + L.off[P_WFC] : sWfc,
                               fc_direct ? c->G + L.off[P_BFC] : sBfc, B, H);
      prof_end(c, ALEPPO_K_FC_WGRAD, sw);
      // bucket 0 = heads + fc.  With data parallelism it is reduced now so that its all-reduce overlaps the conv
      // backward; on one GPU all ten slab groups are reduced by ONE launch after the conv wgrads.
      const ReduceSeg segs0[4] = {{sWh, nblk_head, (long)(A + 1) * H, (long)L.off[P_WH]},
                                  {sBh, nblk_head, (long)A + 1, (long)L.off[P_BH]},
                                  {sWfc, Sfc, (long)H * FC_IN, (long)L.off[P_WFC]},
                                  {sBfc, Sfc, (long)H, (long)L.off[P_BFC]}};
      const int nseg0 = fc_direct ? 2 : 4;
      // bucket 0 is reduced early only for the all-reduce overlap: on one GPU an early reduce next to the conv dgrads
      // measured slower (8.60 vs 8.38 ms per update) than one reduce of all ten slab groups at the end
      const bool early0 = dp;
      if (dp) { // bucket 0 (heads + fc = 95% of the bytes) travels while the conv backward runs.  Its slab reduce runs on
                // the communication stream too, in front of the all-reduce: on the weight-gradient stream it sat between
                // the fc and the conv weight gradients and that stream, not the dgrad chain, ended the minibatch (trace
                // with a 1-rank communicator: conv1 wgrad done at 419 us, conv2 wgrad at 453 us).
        HIPCHK(c, hipEventRecord(c->ev_bucket0, sw));
        HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_bucket0, 0));
        prof_begin(c, ALEPPO_K_REDUCE, c->comm_stream);
        launch_reduce_slabs(c->comm_stream, segs0, nseg0, c->G);
        prof_end(c, ALEPPO_K_REDUCE, c->comm_stream);
        NCCLCHK(c, ncclAllReduce(c->G, c->G, L.bucket0_end, ncclFloat, ncclSum, comm, c->comm_stream));
        HIPCHK(c, hipEventRecord(c->ev_comm0, c->comm_stream));
      }
      HIPCHK(c, fork(c->ev_dz3)); // dz3 is ready
      prof_begin(c, ALEPPO_K_CONV3_DGRAD);
      conv3_dgrad(s, prec, c->dz3, c->W3d, c->a2, c->dz2, B);
      prof_end(c, ALEPPO_K_CONV3_DGRAD);
      prof_begin(c, ALEPPO_K_CONV3_WGRAD, sw);
      const int S3 = conv3_wgrad(sw, prec, c->dz3, c->a2, sW3, sB3, B);
      prof_end(c, ALEPPO_K_CONV3_WGRAD, sw);
      int S2 = 0, S1 = 0, nseg = 0;
      ReduceSeg segs[10];
      if (bwd_fused) {
        // conv2 dgrad + conv2 wgrad + conv1 wgrad in ONE launch on the main stream (conv_bwd_fused.hpp: dz1 never leaves the
        // CU).  Meanwhile the weight-gradient stream reduces the slab groups that are complete (conv3, heads + fc).
        segs[nseg++] = ReduceSeg{sW3, S3, 64 * 576, (long)L.off[P_W3]};
        segs[nseg++] = ReduceSeg{sB3, S3, 64, (long)L.off[P_B3]};
        if (!early0)
          for (int i = 0; i < nseg0; ++i)
            segs[nseg++] = segs0[i];
        if (two) {
          prof_begin(c, ALEPPO_K_REDUCE, sw);
          launch_reduce_slabs(sw, segs, nseg, c->G);
          prof_end(c, ALEPPO_K_REDUCE, sw);
          nseg = 0;
        }
        prof_begin(c, ALEPPO_K_CONV_BWD);
        S2 = S1 = patch_conv_bwd_fused(s, c->dz2, c->a1, c->obs, map, c->W2d, sW2, sB2, sW1, sB1, B);
        prof_end(c, ALEPPO_K_CONV_BWD);
        segs[nseg++] = ReduceSeg{sW2, S2, 64 * 512, (long)L.off[P_W2]};
        segs[nseg++] = ReduceSeg{sB2, S2, 64, (long)L.off[P_B2]};
      } else {
      HIPCHK(c, fork(c->ev_dz2)); // dz2 is ready
      prof_begin(c, ALEPPO_K_CONV2_DGRAD);
      conv2_dgrad(s, prec, c->dz2, c->W2d, c->a1, c->dz1, B);
      prof_end(c, ALEPPO_K_CONV2_DGRAD);
      prof_begin(c, ALEPPO_K_CONV2_WGRAD, sw);
      S2 = conv2_wgrad(sw, prec, c->dz2, c->a1, sW2, sB2, B);
      prof_end(c, ALEPPO_K_CONV2_WGRAD, sw);
      // conv1 wgrad is the last link of the dgrad chain and runs alone on s: meanwhile the wgrad stream reduces every
      // slab group that is already complete (reducing them AFTER conv1 wgrad on the main stream instead measured slower:
      // 7.89-7.96 vs 7.77 ms per update) (conv3, conv2 and - on one GPU - heads + fc); only conv1's slabs are left
      // for the reduce after the join.
      segs[nseg++] = ReduceSeg{sW3, S3, 64 * 576, (long)L.off[P_W3]};
      segs[nseg++] = ReduceSeg{sB3, S3, 64, (long)L.off[P_B3]};
      segs[nseg++] = ReduceSeg{sW2, S2, 64 * 512, (long)L.off[P_W2]};
      segs[nseg++] = ReduceSeg{sB2, S2, 64, (long)L.off[P_B2]};
      if (!early0)
        for (int i = 0; i < nseg0; ++i)
          segs[nseg++] = segs0[i];
      if (two) {
        prof_begin(c, ALEPPO_K_REDUCE, sw);
        launch_reduce_slabs(sw, segs, nseg, c->G);
        prof_end(c, ALEPPO_K_REDUCE, sw);
        nseg = 0;
      }
      prof_begin(c, ALEPPO_K_CONV1_WGRAD);
      S1 = conv1_wgrad(s, prec, c->dz1, c->obs, map, sW1, sB1, B);
      prof_end(c, ALEPPO_K_CONV1_WGRAD);
      }
      if (two) { // join: sumsq / Adam read the whole gradient
        HIPCHK(c, hipEventRecord(c->ev_wg, sw));
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_wg, 0));
      }
      // conv1's slabs: with data parallelism they are reduced now (the all-reduce needs the whole gradient); on one GPU
      // the sum-of-squares pass below sums them on the fly - one launch less on the serial tail of the minibatch.
      ReduceSeg tail[2] = {{sW1, S1, 32 * 256, (long)L.off[P_W1]}, {sB1, S1, 32, (long)L.off[P_B1]}};
      const bool fuse_tail = fuse_tail_env();
      if (dp || !fuse_tail) {
        segs[nseg++] = tail[0];
        segs[nseg++] = tail[1];
        tail[0].slab = tail[1].slab = nullptr;
      }
      if (nseg) {
        prof_begin(c, ALEPPO_K_REDUCE);
        launch_reduce_slabs(s, segs, nseg, c->G);
        prof_end(c, ALEPPO_K_REDUCE);
      }
      if (dp) {
        // Bucket 1 (the conv tensors, 0.35 MB) is on the critical path whatever stream carries it - nothing is left to
        // overlap it with - so it runs on the MAIN stream: a round trip through the communication stream cost two more
        // cross-stream dependencies (~12 us each, forkbench) in the serial tail of every minibatch (update with a 1-rank
        // communicator: +31 -> +16 us per minibatch over the single-GPU schedule).  The wait for bucket 0's event comes
        // first: two collectives of one communicator must never be in flight together.
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_comm0, 0));
        NCCLCHK(c, ncclAllReduce(c->G + L.bucket0_end, c->G + L.bucket0_end, L.total() - L.bucket0_end, ncclFloat,
                                 ncclSum, comm, s));
      }
      prof_begin(c, ALEPPO_K_ADAM);
      // (pads between tensors are zero: only [0, off[P_W1]) and the two conv1 tensors contribute)
      const int nblk_norm = launch_sumsq(s, c->G, (long)L.off[P_W1], c->sumsq_part, nblk_sq, tail);
      // (the Adam kernel also writes the bf16 compute copy and the dgrad-side transposed layouts W2d / W3d / WfcT)
      launch_adam(s, c->P, c->G, nullptr, c->M1, c->M2, c->prec == ALEPPO_BF16 ? c->Pc : nullptr, c->WfcT, c->W3d, c->W2d,
                  L, prec, c->sumsq_part, nblk_norm, hp.max_norm, c->adam_sched + 2 * mi, c->cfg.adam_beta1,
                  c->cfg.adam_beta2, c->cfg.adam_eps, c->grad_norms + mi);
      prof_end(c, ALEPPO_K_ADAM);
    }
  launch_metrics_reduce(s, c->metric_ps, fs, c->mask_n, B, M, epochs, c->metric_red);
  if (dp)
    NCCLCHK(c, ncclAllReduce(c->metric_red, c->metric_red, (size_t)nm * 8, ncclFloat, ncclSum, comm, s));
  return ALEPPO_OK;
  }; // enqueue_update

  // ALEPPO_OPT_UPDATE_GRAPH (capture_train_cuda_graph, train.h:163-195): the first call of a shape runs eagerly (it also
  // performs the kernels' one-time attribute set-up), the second records the same enqueue into a graph, later calls
  // replay it.  Not with data parallelism (the collectives stay eager) and not while per-kernel profiling brackets launches.
  Ctx::GraphKey key;
  key.epochs = epochs;
  key.M = M;
  key.two = two ? 1 : 0;
  key.N = N;
  key.metric_ps = c->metric_ps;
  key.metric_red = c->metric_red;
  const bool want_graph = c->update_graph && !dp && !c->prof_on;
  if (want_graph && c->graph_exec && c->graph_key == key) {
    HIPCHK(c, hipGraphLaunch(c->graph_exec, s));
    c->graph_replays++;
  } else if (want_graph && c->warm_key == key) {
    if (c->graph_exec)
      HIPCHK(c, hipGraphExecDestroy(c->graph_exec));
    if (c->graph)
      HIPCHK(c, hipGraphDestroy(c->graph));
    c->graph_exec = nullptr;
    c->graph = nullptr;
    HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    rc = enqueue_update();
    hipGraph_t g = nullptr;
    hipError_t ee = hipStreamEndCapture(s, &g); // (also ends a capture that failed half way)
    if (rc == ALEPPO_OK && ee == hipSuccess)
      ee = hipGraphInstantiate(&c->graph_exec, g, nullptr, nullptr, 0);
    if (rc || ee != hipSuccess) {
      // Nothing has run yet (a capture only records).  Drop the half-built graph and its keys so that the next call
      // starts from the eager path again instead of re-capturing for ever, and report.
      if (g)
        hipGraphDestroy(g);
      c->graph_exec = nullptr;
      c->graph_key = Ctx::GraphKey();
      c->warm_key = Ctx::GraphKey();
      if (rc)
        return rc;
      HIPCHK(c, ee);
    }
    c->graph = g;
    c->graph_key = key;
    HIPCHK(c, hipGraphLaunch(c->graph_exec, s));
    c->graph_replays++;
  } else {
    rc = enqueue_update();
    if (rc) // some optimizer steps may already be on the stream: parameters / Adam state are no longer what the caller
            // thinks they are, and adam_step cannot say how far the device got
      return fail_ctx(c, rc, "aleppo_train failed while enqueuing the update (" + c->err + ")");
    c->warm_key = key;
  }
  c->adam_step += nm;
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_metric_red, c->metric_red, (size_t)nm * 8 * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipMemcpyAsync(c->h_metric_red + (size_t)nm * 8, c->grad_norms, (size_t)nm * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipStreamSynchronize(s));
  CHECK_ASYNC(c);
  c->last_epochs = epochs;
  c->last_M = M;
  c->last_B = B;
  if (out)
    for (int i = 0; i < nm; ++i) {
      const float *r = c->h_metric_red + (size_t)i * 8;
      const float cnt = r[5];
      out[i].loss = r[0] / cnt;
      out[i].clipped_loss = r[1] / cnt;
      out[i].value_loss = r[2] / cnt;
      out[i].entropy = r[3] / cnt;
      out[i].ratio = r[4] / cnt;
      out[i].mask_count = cnt;
      out[i].grad_norm = c->h_metric_red[(size_t)nm * 8 + i];
    }
  return ALEPPO_OK;
}

extern "C" int aleppo_read_train_metric(aleppo_ctx *c, int field, float *dst, size_t count) {
  CHECK_CTX(c);
  const size_t n = (size_t)c->last_epochs * c->last_M * c->last_B;
  if (!dst || field < 0 || field > 4 || count != n || n == 0)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "read_train_metric: bad field or count");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, copy_sync(c, dst, c->metric_ps + (size_t)field * c->metric_cap, n * 4, hipMemcpyDeviceToHost));
  return ALEPPO_OK;
}

// NCHW uint8 observations of the caller -> c->stage_u8 (device), grown on demand and kept
static int stage_observations(aleppo_ctx *c, const uint8_t *observations, int64_t n) {
  const size_t bytes = (size_t)n * 4 * FRAME_PIX;
  if (bytes > c->stage_u8_cap) {
    retire(c, c->stage_u8);
    c->stage_u8 = nullptr;
    c->stage_u8_cap = 0;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->stage_u8), bytes));
    c->stage_u8_cap = bytes;
  }
  HIPCHK(c, copy_sync(c, c->stage_u8, observations, bytes, hipMemcpyHostToDevice));
  return ALEPPO_OK;
}

extern "C" int aleppo_set_batch(aleppo_ctx *c, const uint8_t *observations, const int64_t *actions,
                                const float *log_probabilities, const float *advantages, const float *returns,
                                const uint8_t *masks, int64_t n) {
  CHECK_CTX(c);
  if (!observations || !actions || !log_probabilities || !advantages || !returns || !masks)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  if (n <= 0 || n > c->N)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "set_batch: n must be in [1, E*T]");
  std::vector<int> a32(n);
  for (int64_t i = 0; i < n; ++i) {
    if (actions[i] < 0 || actions[i] >= c->A)
      return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "action index out of range");
    a32[i] = (int)actions[i];
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->pre_acted = -1;
  int rc = stage_observations(c, observations, n);
  if (rc)
    return rc;
  launch_obs_pack(c->stream, c->stage_u8, c->obs, n, train_map(c, 0));
  HIPCHK(c, copy_sync(c, c->act_n, a32.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  if (!c->rt16) {
    HIPCHK(c, copy_sync(c, c->oldlp_n, log_probabilities, (size_t)n * c->A * 4, hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->adv_n, advantages, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, copy_sync(c, c->ret_n, returns, (size_t)n * 4, hipMemcpyHostToDevice));
  } else { // half planes: upload as float into the (idle) metric scratch area, round on the device
    rc = ensure_metric_storage(c, 1, 1, (long)n * std::max(c->A, 1));
    if (rc)
      return rc;
    const struct {
      const float *src;
      void *dst;
      size_t cnt;
    } pl[3] = {{log_probabilities, c->oldlp_n, (size_t)n * c->A}, {advantages, c->adv_n, (size_t)n}, {returns, c->ret_n, (size_t)n}};
    for (const auto &q : pl) {
      HIPCHK(c, copy_sync(c, c->metric_ps, q.src, q.cnt * 4, hipMemcpyHostToDevice));
      launch_plane_from_float(c->stream, c->metric_ps, q.dst, (long)q.cnt, true);
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
  }
  HIPCHK(c, copy_sync(c, c->mask_n, masks, (size_t)n, hipMemcpyHostToDevice));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->batch_n = n;
  return ALEPPO_OK;
}

extern "C" int aleppo_forward(aleppo_ctx *c, const uint8_t *observations, int64_t n, float *logits, float *values) {
  CHECK_CTX(c);
  if (!observations || !logits || !values)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null argument");
  if (n <= 0 || n > c->maxB)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "forward: n exceeds capacity");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->pre_acted = -1; // a3 / h are shared scratch
  int rc = stage_observations(c, observations, n);
  if (rc)
    return rc;
  // the packed stacks go to a staging area of their own: the rollout's observation slots are not touched
  const size_t need = (size_t)n * FRAME_PIX * 4;
  if (need > c->stage_obs_cap) {
    retire(c, c->stage_obs);
    c->stage_obs = nullptr;
    c->stage_obs_cap = 0;
    HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&c->stage_obs), need));
    c->stage_obs_cap = need;
  }
  const SampleMap map{1, (long)FRAME_PIX, 0, 0, 0}; // sample n at stage_obs + n * 7056
  launch_obs_pack(c->stream, c->stage_u8, c->stage_obs, n, map);
  net_forward(c, c->stage_obs, map, n);
  launch_heads_fwd(c->stream, c->h, Pf(c, P_WH), Pf(c, P_BH), c->logits_b, c->values_b, n, c->H, c->A);
  HIPCHK(c, hipMemcpyAsync(logits, c->logits_b, (size_t)n * c->A * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(values, c->values_b, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  CHECK_ASYNC(c);
  return ALEPPO_OK;
}

extern "C" int aleppo_read_batch(aleppo_ctx *c, int field, void *dst, size_t bytes) {
  CHECK_CTX(c);
  if (!dst)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null dst");
  const int E = c->E, T = c->T, A = c->A;
  const size_t N = (size_t)c->N;
  hipStream_t s = c->stream;
  size_t need = 0;
  // device scratch owned by the context, grown on demand and never freed before aleppo_destroy (hipFree would wait for
  // every stream of the device, other contexts' parked ones included: see dalloc)
  auto scratch = [&](int k, size_t nbytes, void **out) -> int {
    if (nbytes > c->rb_cap[k]) {
      retire(c, c->rb_tmp[k]);
      c->rb_tmp[k] = nullptr;
      c->rb_cap[k] = 0;
      HIPCHK(c, hipMalloc(&c->rb_tmp[k], nbytes));
      c->rb_cap[k] = nbytes;
    }
    *out = c->rb_tmp[k];
    return ALEPPO_OK;
  };
  void *tmp = nullptr;
  auto fin = [&](const void *src) -> int {
    if (bytes != need)
      return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "read_batch: wrong byte count");
    HIPCHK(c, copy_sync(c, dst, src, need, hipMemcpyDeviceToHost));
    return ALEPPO_OK;
  };
  auto transposed = [&](const void *src, size_t pitch, int inner, int elem) -> int {
    need = N * inner * elem;
    if (int rc = scratch(0, need, &tmp))
      return rc;
    launch_transpose_tm_pitched(s, src, pitch, tmp, E, T, inner, elem);
    return fin(tmp);
  };
  // float planes stored as RT: [count] elements, env-major already (tm = false) or time-major [T][E][inner]
  auto plane = [&](const void *src, size_t count, bool tm, int inner) -> int {
    need = count * 4;
    if (bytes != need)
      return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "read_batch: wrong byte count");
    if (!c->rt16 && !tm)
      return fin(src);
    if (int rc = scratch(0, need, &tmp))
      return rc;
    if (tm) {
      void *em = nullptr;
      if (int rc = scratch(1, count * c->rsz, &em))
        return rc;
      launch_transpose_tm_pitched(s, src, (size_t)E * inner * c->rsz, em, E, T, inner, (int)c->rsz);
      src = em;
    }
    launch_plane_to_float(s, src, static_cast<float *>(tmp), (long)count, c->rt16);
    return fin(tmp);
  };
  switch (field) {
  case ALEPPO_F_OBSERVATIONS: {
    need = N * 4 * FRAME_PIX;
    if (int rc = scratch(0, need, &tmp))
      return rc;
    launch_obs_unpack(s, c->obs, static_cast<uint8_t *>(tmp), (long)N, train_map(c, 0));
    return fin(tmp);
  }
  case ALEPPO_F_CURRENT_OBS: {
    need = (size_t)E * 4 * FRAME_PIX;
    if (int rc = scratch(0, need, &tmp))
      return rc;
    const int slot = (c->t == 0 && c->need_carry) ? T : c->t;
    launch_obs_unpack(s, c->obs, static_cast<uint8_t *>(tmp), E, slot_map(c, slot));
    return fin(tmp);
  }
  case ALEPPO_F_ACTIONS: {
    need = N * 8;
    if (bytes != need)
      return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "read_batch: wrong byte count");
    std::vector<int> a(N);
    HIPCHK(c, copy_sync(c, a.data(), c->act_n, N * 4, hipMemcpyDeviceToHost));
    int64_t *o = static_cast<int64_t *>(dst);
    for (size_t i = 0; i < N; ++i)
      o[i] = a[i];
    return ALEPPO_OK;
  }
  case ALEPPO_F_REWARDS:
    return transposed(c->step_rec, c->step_rec_bytes, 1, 4);
  case ALEPPO_F_TERMINALS:
    return transposed(c->step_rec + 4 * (size_t)E, c->step_rec_bytes, 1, 1);
  case ALEPPO_F_TRUNCATIONS:
    return transposed(c->step_rec + 5 * (size_t)E, c->step_rec_bytes, 1, 1);
  case ALEPPO_F_LOGITS:
    return plane(c->logits_tm, N * A, true, A);
  case ALEPPO_F_VALUES:
    return plane(c->values_tm, N, true, 1);
  case ALEPPO_F_MASKS:
    need = N;
    return fin(c->mask_n);
  case ALEPPO_F_ADVANTAGES:
    return plane(c->adv_n, N, false, 1);
  case ALEPPO_F_RETURNS:
    return plane(c->ret_n, N, false, 1);
  case ALEPPO_F_LOG_PROBS:
    return plane(c->oldlp_n, N * A, false, A);
  case ALEPPO_F_NEXT_VALUES:
    return plane(rp(c, c->values_tm, (size_t)T * E), (size_t)E, false, 1);
  default:
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "read_batch: unknown field");
  }
}

// ------------------------------------------------------------------ multi-GPU
extern "C" int aleppo_comm_unique_id(uint8_t id[ALEPPO_UNIQUE_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == ALEPPO_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess)
    return set_err(nullptr, ALEPPO_ERR_HIP, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
  std::memcpy(id, &u, sizeof(u));
  return ALEPPO_OK;
}
extern "C" int aleppo_comm_init(aleppo_ctx *c, const uint8_t id[ALEPPO_UNIQUE_ID_BYTES]) {
  CHECK_CTX(c);
  if (c->nccl_comm)
    return set_err(c, ALEPPO_ERR_RUNTIME, "communicator already initialised");
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  HIPCHK(c, hipSetDevice(c->cfg.device_ordinal));
  ncclComm_t comm;
  NCCLCHK(c, ncclCommInitRank(&comm, c->world, u, c->rank));
  c->nccl_comm = comm;
  return ALEPPO_OK;
}

// ------------------------------------------------------------------ profiling
extern "C" int aleppo_set_option(aleppo_ctx *c, int option, int value) {
  CHECK_CTX(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // a captured update holds the kernels the switches selected when it was recorded: any change re-arms the capture
  c->graph_key = Ctx::GraphKey();
  c->warm_key = Ctx::GraphKey();
  if (option == ALEPPO_OPT_GENERIC_CONV)
    c->tune.patch_conv = value == 0;
  else if (option == ALEPPO_OPT_FC_PIPE)
    c->tune.fc_pipe = value != 0;
  else if (option == ALEPPO_OPT_FUSED_ACT)
    c->tune.fused_act = value; // 0: never, 1: where it is faster (default), 2: always
  else if (option == ALEPPO_OPT_FUSED_FWD)
    c->tune.fused_fwd = value != 0;
  else if (option == ALEPPO_OPT_FUSED_BWD)
    c->tune.fused_bwd = value; // 0: never, 1: at minibatches >= 2048 samples (default), 2: always
  else if (option == ALEPPO_OPT_DEBUG_NO_PUBLISH)
    c->dbg_no_publish = value != 0;
  else if (option == ALEPPO_OPT_SERIAL_UPDATE)
    c->serial_update = value != 0;
  else if (option == ALEPPO_OPT_FORCE_COMM)
    c->force_comm = value != 0;
  else if (option == ALEPPO_OPT_UPDATE_GRAPH)
    c->update_graph = value != 0;
  else if (option == ALEPPO_OPT_GATE_TIMEOUT_MS)
    c->gate_timeout_ticks = (unsigned long long)std::max(1, value) * 100000ull; // 100 MHz wall clock
  else
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "unknown option");
  return ALEPPO_OK;
}
extern "C" int aleppo_get_option(aleppo_ctx *c, int option, int64_t *value) {
  CHECK_CTX(c);
  if (!value)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "null value");
  switch (option) {
  case ALEPPO_OPT_GENERIC_CONV: *value = c->tune.patch_conv ? 0 : 1; break;
  case ALEPPO_OPT_DEBUG_NO_PUBLISH: *value = c->dbg_no_publish; break;
  case ALEPPO_OPT_FORCE_COMM: *value = c->force_comm; break;
  case ALEPPO_OPT_SERIAL_UPDATE: *value = c->serial_update; break;
  case ALEPPO_OPT_FC_PIPE: *value = c->tune.fc_pipe; break;
  case ALEPPO_OPT_FUSED_ACT: *value = c->tune.fused_act; break;
  case ALEPPO_OPT_FUSED_FWD: *value = c->tune.fused_fwd; break;
  case ALEPPO_OPT_FUSED_BWD: *value = c->tune.fused_bwd; break;
  case ALEPPO_OPT_UPDATE_GRAPH: *value = c->graph_replays; break;
  case ALEPPO_OPT_GATE_TIMEOUT_MS: *value = (int64_t)(c->gate_timeout_ticks / 100000ull); break;
  default: return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "unknown option");
  }
  return ALEPPO_OK;
}
extern "C" int aleppo_profile_enable(aleppo_ctx *c, int on) {
  CHECK_CTX(c);
  c->prof_on = on != 0;
  return ALEPPO_OK;
}
extern "C" int aleppo_profile_reset(aleppo_ctx *c) {
  CHECK_CTX(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto &p : c->prof)
    p.used = 0;
  return ALEPPO_OK;
}
extern "C" int aleppo_profile_read(aleppo_ctx *c, int cls, double *avg_ms, int64_t *launches) {
  CHECK_CTX(c);
  if (cls < 0 || cls >= ALEPPO_K_COUNT || !avg_ms || !launches)
    return set_err(c, ALEPPO_ERR_INVALID_ARGUMENT, "profile_read: bad argument");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  ProfClass &p = c->prof[cls];
  double tot = 0;
  for (size_t i = 0; i < p.used; ++i) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
    tot += ms;
  }
  *launches = (int64_t)p.used;
  *avg_ms = p.used ? tot / (double)p.used : 0.0;
  return ALEPPO_OK;
}

// ------------------------------------------------------------------ stateless operators
// Device memory and the stream of the stateless operators: a process-wide arena per device, grown in chunks that are
// never freed, and a non-blocking stream of its own.  (hipFree, hipDeviceSynchronize and null-stream work would wait for,
// or order themselves against, every other stream of the device - a context's stream parked behind its release word
// included)  One operator call at a time per device (mutex); a call's buffers live until it returns.
namespace {
struct OpArena {
  std::mutex mu;
  hipStream_t st = nullptr;
  struct Chunk {
    char *p;
    size_t cap, used;
  };
  std::vector<Chunk> chunks;
};
OpArena &op_arena(int dev) {
  static OpArena a[64];
  return a[dev & 63];
}
struct OpScope;
thread_local OpScope *g_op = nullptr;
struct OpScope {
  OpArena &ar;
  std::unique_lock<std::mutex> lk;
  hipStream_t st = nullptr;
  hipError_t err = hipSuccess;
  explicit OpScope(int dev) : ar(op_arena(dev)), lk(ar.mu) {
    if (!ar.st)
      err = hipStreamCreateWithFlags(&ar.st, hipStreamNonBlocking);
    st = ar.st;
    for (auto &ch : ar.chunks)
      ch.used = 0;
    g_op = this;
  }
  ~OpScope() { g_op = nullptr; }
  hipError_t alloc(void **out, size_t bytes) {
    bytes = (std::max<size_t>(bytes, 16) + 255) / 256 * 256;
    for (auto &ch : ar.chunks)
      if (ch.cap - ch.used >= bytes) {
        *out = ch.p + ch.used;
        ch.used += bytes;
        return hipSuccess;
      }
    OpArena::Chunk ch{nullptr, std::max<size_t>(bytes, (size_t)32 << 20), bytes};
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(&ch.p), ch.cap);
    if (e != hipSuccess)
      return e;
    ar.chunks.push_back(ch);
    *out = ch.p;
    return hipSuccess;
  }
  hipError_t sync() { return hipStreamSynchronize(st); }
  hipError_t down(void *dst, const void *src, size_t bytes) { // device -> host, complete on return
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    return e == hipSuccess ? sync() : e;
  }
};
struct DevBuf { // a buffer of the current operator call (arena memory: nothing to free)
  void *p = nullptr;
  hipError_t up(const void *src, size_t bytes) {
    hipError_t e = g_op->alloc(&p, bytes);
    if (e == hipSuccess && src)
      e = hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, g_op->st);
    else if (e == hipSuccess)
      e = hipMemsetAsync(p, 0, bytes ? bytes : 16, g_op->st);
    // (pageable host sources are staged before the call returns; waiting here keeps the caller's buffer rule simple)
    return e == hipSuccess ? g_op->sync() : e;
  }
  template <class T> T *as() { return static_cast<T *>(p); }
};
} // namespace
#define OPCHK(x)                                                                                                       \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess)                                                                                              \
      return set_err(nullptr, ALEPPO_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                         \
  } while (0)

// ai::gae::gae through the production scan kernel: the host arrays are laid out as the rollout's time-major step
// records + value plane (the layout aleppo_finish_rollout hands the kernel), gae_kernel runs with clamp = 0 (ai::gae::gae
// does not clamp; Buffer::get does, buffer.cc:67) and the env-major advantage array comes back as is.
extern "C" int aleppo_gae(int dev, float *advantages, const float *rewards, const float *values,
                          const float *next_values, const uint8_t *terminals, const uint8_t *truncations,
                          const uint8_t *episode_starts, int64_t E, int64_t T, float gamma, float lambda) {
  if (!advantages || !rewards || !values || !next_values || !terminals || !truncations || !episode_starts)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT,
                   "All input tensors must be 2D except next_values which must be 1D."); // gae.cc:8-13
  if (E <= 0 || T <= 0)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "Input tensors must have compatible dimensions."); // :14-21
  int rc = select_device(dev);
  if (rc)
    return rc;
  OpScope op(dev);
  OPCHK(op.err);
  const size_t n = (size_t)E * T, rb = ((size_t)7 * E + 15) / 16 * 16;
  std::vector<uint8_t> rec(rb * T, 0);
  std::vector<float> vtm((size_t)(T + 1) * E);
  for (int64_t t = 0; t < T; ++t) {
    uint8_t *r = rec.data() + (size_t)t * rb;
    for (int64_t e = 0; e < E; ++e) {
      reinterpret_cast<float *>(r)[e] = rewards[e * T + t];
      r[4 * E + e] = terminals[e * T + t];
      r[5 * E + e] = truncations[e * T + t];
      r[6 * E + e] = episode_starts[e * T + t];
      vtm[(size_t)t * E + e] = values[e * T + t];
    }
  }
  for (int64_t e = 0; e < E; ++e)
    vtm[(size_t)T * E + e] = next_values[e];
  DevBuf drec, dv, a, r, m, er;
  OPCHK(drec.up(rec.data(), rec.size()));
  OPCHK(dv.up(vtm.data(), vtm.size() * 4));
  OPCHK(a.up(nullptr, n * 4));
  OPCHK(r.up(nullptr, n * 4));
  OPCHK(m.up(nullptr, n));
  OPCHK(er.up(nullptr, 16));
  launch_gae(op.st, drec.as<uint8_t>(), rb, dv.as<float>(), nullptr, nullptr, a.as<float>(), r.as<float>(), nullptr,
             nullptr, m.as<uint8_t>(), er.as<int>(), (int)E, (int)T, 0, gamma, lambda, /*clamp=*/false);
  int err = 0;
  OPCHK(op.down(&err, er.p, 4));
  if (err)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT,
                   "Episode starts, terminals, and truncations must be mutually exclusive."); // gae.cc:49-53
  OPCHK(op.down(advantages, a.p, n * 4));
  return ALEPPO_OK;
}

extern "C" int aleppo_vision_resize_area(int dev, const float *images, float *out, int64_t n) {
  if (!images || !out || n <= 0)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "bad argument");
  int rc = select_device(dev);
  if (rc)
    return rc;
  OpScope op(dev);
  OPCHK(op.err);
  DevBuf i, o;
  OPCHK(i.up(images, (size_t)n * RAW_H * RAW_W * 4));
  OPCHK(o.up(nullptr, (size_t)n * FRAME_PIX * 4));
  launch_area_resize(op.st, i.as<float>(), o.as<float>(), n);
  OPCHK(op.down(out, o.p, (size_t)n * FRAME_PIX * 4));
  return ALEPPO_OK;
}
extern "C" int aleppo_vision_rgb_to_gray(int dev, const float *images, float *out, int64_t n) {
  if (!images || !out || n <= 0)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "bad argument");
  int rc = select_device(dev);
  if (rc)
    return rc;
  OpScope op(dev);
  OPCHK(op.err);
  DevBuf i, o;
  OPCHK(i.up(images, (size_t)n * 3 * FRAME_PIX * 4));
  OPCHK(o.up(nullptr, (size_t)n * FRAME_PIX * 4));
  launch_rgb_to_gray(op.st, i.as<float>(), o.as<float>(), n);
  OPCHK(op.down(out, o.p, (size_t)n * FRAME_PIX * 4));
  return ALEPPO_OK;
}

// The two frame operators run the production ingest kernel on a scratch pair of observation slots
// ([n][2 slots][7056] packed stacks, slot 0 = before, slot 1 = after) and convert at the boundary.
static int ingest_op(bool raw, const uint8_t *frames, size_t frame_bytes, const uint8_t *lut256,
                     const uint8_t *obs_nchw_in, const uint8_t *start, uint8_t *obs_nchw_out, int64_t n) {
  if (n > MAX_ENVS_PER_RANK)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "at most 8192 environments per call");
  OpScope &op = *g_op; // opened by the caller
  DevBuf f, l, st, nchw, slots;
  OPCHK(f.up(frames, frame_bytes));
  uint8_t ident[256];
  for (int i = 0; i < 256; ++i)
    ident[i] = (uint8_t)i;
  OPCHK(l.up(lut256 ? lut256 : ident, 256));
  std::vector<uint8_t> ones;
  if (!start) {
    ones.assign((size_t)n, 1);
    start = ones.data();
  }
  OPCHK(st.up(start, (size_t)n));
  OPCHK(nchw.up(obs_nchw_in, (size_t)n * 4 * FRAME_PIX)); // (zeros when there is no previous stack)
  OPCHK(slots.up(nullptr, (size_t)n * 2 * FRAME_PIX * 4));
  launch_obs_pack(op.st, nchw.as<uint8_t>(), slots.as<uint32_t>(), n, SampleMap{1, 2L * FRAME_PIX, 0, 0, 0});
  launch_ingest(op.st, raw, f.as<uint8_t>(), l.as<uint8_t>(), st.as<uint8_t>(), nullptr, slots.as<uint32_t>(), (int)n,
                2, 0, 1);
  launch_obs_unpack(op.st, slots.as<uint32_t>(), nchw.as<uint8_t>(), n, SampleMap{1, 2L * FRAME_PIX, 0, FRAME_PIX, 0});
  OPCHK(op.sync());
  OPCHK(op.down(obs_nchw_out, nchw.p, (size_t)n * 4 * FRAME_PIX));
  return ALEPPO_OK;
}
extern "C" int aleppo_preprocess(int dev, const uint8_t *raw_pairs, const uint8_t *lut256, uint8_t *out, int64_t n) {
  if (!raw_pairs || !out || n <= 0)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "bad argument");
  int rc = select_device(dev);
  if (rc)
    return rc;
  OpScope op(dev);
  OPCHK(op.err);
  // every environment in an episode-start slot: the new frame is broadcast to all four stack planes; plane 0 is it
  std::vector<uint8_t> stack((size_t)n * 4 * FRAME_PIX);
  rc = ingest_op(true, raw_pairs, (size_t)n * 2 * RAW_H * RAW_W, lut256, nullptr, nullptr, stack.data(), n);
  if (rc)
    return rc;
  for (int64_t e = 0; e < n; ++e)
    std::memcpy(out + (size_t)e * FRAME_PIX, stack.data() + (size_t)e * 4 * FRAME_PIX, FRAME_PIX);
  return ALEPPO_OK;
}
extern "C" int aleppo_update_observations(int dev, uint8_t *observations, const uint8_t *frames,
                                          const uint8_t *episode_start, int64_t E) {
  if (!observations || !frames || !episode_start || E <= 0)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "bad argument");
  int rc = select_device(dev);
  if (rc)
    return rc;
  OpScope op(dev);
  OPCHK(op.err);
  return ingest_op(false, frames, (size_t)E * FRAME_PIX, nullptr, observations, episode_start, observations, E);
}

// ai::ppo::losses::compute through the production head kernel (head_train_kernel<float>): the caller's raw logits and
// values become the first A + 1 components of a 32-wide hidden vector and the head weights an identity block, so the
// kernel's "head linear layer" reproduces them exactly (x * 1 + 0 + ... is exact in fp32) and its dh output IS
// (dlogits, dvalue).  The scalar loss is the masked mean the update reports (metrics_reduce_kernel, as aleppo_train).
extern "C" int aleppo_ppo_loss(int dev, const float *logits, const float *old_lp, const int64_t *actions,
                               const float *advantages, const float *values, const float *returns,
                               const uint8_t *masks, int64_t B, int64_t A, float clip, float c_v, float c_e,
                               float *loss, float *clipped, float *value_losses, float *entropies, float *total_losses,
                               float *ratio, float *dlogits, float *dvalues) {
  if (!logits || !old_lp || !actions || !advantages || !values || !returns || !masks || B <= 0 || A <= 0 ||
      A > MAX_ACTIONS)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "bad argument");
  std::vector<int> a32((size_t)B);
  for (int64_t i = 0; i < B; ++i) {
    if (actions[i] < 0 || actions[i] >= A)
      return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "action index out of range");
    a32[(size_t)i] = (int)actions[i];
  }
  int rc = select_device(dev);
  if (rc)
    return rc;
  OpScope op(dev);
  OPCHK(op.err);
  constexpr int H = 32; // >= MAX_ACTIONS + 1
  std::vector<float> h((size_t)B * H, 0.f), Wh((size_t)(A + 1) * H, 0.f), bh((size_t)A + 1, 0.f);
  for (int64_t i = 0; i < B; ++i) {
    for (int64_t k = 0; k < A; ++k)
      h[(size_t)i * H + k] = logits[i * A + k];
    h[(size_t)i * H + A] = values[i];
  }
  for (int64_t k = 0; k <= A; ++k)
    Wh[(size_t)k * H + k] = 1.0f;
  const int nblk = (int)std::min<int64_t>(MAXS_HEAD, (B + 15) / 16);
  DevBuf dh_in, dW, db, ol, ac, ad, re, ma, cnt, dh_out, ps, sw, sb, red;
  OPCHK(dh_in.up(h.data(), h.size() * 4));
  OPCHK(dW.up(Wh.data(), Wh.size() * 4));
  OPCHK(db.up(bh.data(), bh.size() * 4));
  OPCHK(ol.up(old_lp, (size_t)B * A * 4));
  OPCHK(ac.up(a32.data(), (size_t)B * 4));
  OPCHK(ad.up(advantages, (size_t)B * 4));
  OPCHK(re.up(returns, (size_t)B * 4));
  OPCHK(ma.up(masks, (size_t)B));
  OPCHK(cnt.up(nullptr, 16));
  OPCHK(dh_out.up(nullptr, (size_t)B * H * 4));
  OPCHK(ps.up(nullptr, (size_t)5 * B * 4));
  OPCHK(sw.up(nullptr, (size_t)nblk * (A + 1) * H * 4));
  OPCHK(sb.up(nullptr, (size_t)nblk * (A + 1) * 4));
  OPCHK(red.up(nullptr, 8 * 4));
  launch_mask_count(op.st, ma.as<uint8_t>(), cnt.as<float>(), B, 1); // losses.cc:19 masks.sum()
  float *p = ps.as<float>();
  launch_head_train(op.st, dh_in.as<float>(), dW.as<float>(), db.as<float>(), ac.as<int>(), ol.as<float>(),
                    ad.as<float>(), re.as<float>(), ma.as<uint8_t>(), cnt.as<float>(), Hyper{clip, c_v, c_e, 0.f},
                    dh_out.p, ALEPPO_FP32, p, p + B, p + 2 * B, p + 3 * B, p + 4 * B, sw.as<float>(), sb.as<float>(),
                    nblk, B, H, (int)A, nullptr, nullptr, 1);
  launch_metrics_reduce(op.st, p, (size_t)B, ma.as<uint8_t>(), B, 1, 1, red.as<float>());
  OPCHK(op.sync());
  float r8[8];
  OPCHK(op.down(r8, red.p, sizeof(r8)));
  if (loss)
    *loss = r8[0] / r8[5];
  float *per[5] = {total_losses, clipped, value_losses, entropies, ratio}; // order of the kernel's metric planes
  for (int k = 0; k < 5; ++k)
    if (per[k])
      OPCHK(op.down(per[k], p + (size_t)k * B, (size_t)B * 4));
  if (dlogits || dvalues) {
    std::vector<float> d((size_t)B * H);
    OPCHK(op.down(d.data(), dh_out.p, d.size() * 4));
    for (int64_t i = 0; i < B; ++i) {
      if (dlogits)
        for (int64_t k = 0; k < A; ++k)
          dlogits[i * A + k] = d[(size_t)i * H + k];
      if (dvalues)
        dvalues[i] = d[(size_t)i * H + A];
    }
  }
  return ALEPPO_OK;
}
// multinomial(probs, 1, true) given its noise through the production acting head (infer_head_kernel in its probs mode:
// same division, same wave arg-max, same stores)
extern "C" int aleppo_sample(int dev, const float *probs, const float *q, int64_t *actions, int64_t E, int64_t A) {
  if (!probs || !q || !actions || E <= 0 || A <= 0 || A > MAX_ACTIONS)
    return set_err(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "bad argument");
  int rc = select_device(dev);
  if (rc)
    return rc;
  OpScope op(dev);
  OPCHK(op.err);
  DevBuf p, qq, a;
  OPCHK(p.up(probs, (size_t)E * A * 4));
  OPCHK(qq.up(q, (size_t)E * A * 4));
  OPCHK(a.up(nullptr, (size_t)E * 4));
  launch_infer_head(op.st, nullptr, FC_SPLITS, nullptr, nullptr, nullptr, qq.as<float>(), 0, 0, nullptr, nullptr,
                    a.as<int>(), nullptr, nullptr, 0, (int)E, 32, (int)A, p.as<float>());
  std::vector<int> a32((size_t)E);
  OPCHK(op.down(a32.data(), a.p, (size_t)E * 4));
  for (int64_t e = 0; e < E; ++e)
    actions[e] = a32[(size_t)e];
  return ALEPPO_OK;
}
