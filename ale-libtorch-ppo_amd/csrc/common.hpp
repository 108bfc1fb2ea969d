// common.hpp - internal context, HBM layout and launcher declarations of libaleppo.so.
// Public boundary: include/aleppo.h.  Layout rationale: DESIGN.md.
#pragma once
#include "../../include/aleppo.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace aleppo {

constexpr int FRAME_PIX = 84 * 84;          // 7056 packed pixels per stack (one u32 = 4 frames of one pixel)
constexpr int RAW_H = 210, RAW_W = 160;
constexpr int A1_PIX = 400, A1_C = 32;      // conv1 out 20x20x32 (NHWC)
constexpr int A2_PIX = 81, A2_C = 64;       // conv2 out 9x9x64
constexpr int A3_PIX = 49, A3_C = 64;       // conv3 out 7x7x64
constexpr int FC_IN = 3136;
constexpr int MAX_ACTIONS = 18;
constexpr int FC_SPLITS = 7; // split-K slices of the acting-size fc forward (3136 = 7 * 448)
// split-K slice caps of the wgrad slabs (gemm_launch.hip) and of the head kernel's partial slabs
constexpr int MAXS_C1 = 256, MAXS_C2 = 256, MAXS_C3 = 256, MAXS_FC = 8, MAXS_HEAD = 256;

// ---- internal flat parameter layout (fp32 master, Adam moments, gradient share it) ----
// order chosen so that what backward finishes FIRST is at the FRONT: bucket 0 = heads + fc can be
// all-reduced while the conv backward still runs.  Every tensor starts 64-float aligned;
// pads are zero and stay zero under Adam.
enum ParamId { P_WH = 0, P_BH, P_WFC, P_BFC, P_W3, P_B3, P_W2, P_B2, P_W1, P_B1, P_COUNT };
struct ParamLayout {
  size_t off[P_COUNT + 1]; // off[P_COUNT] = total (padded)
  size_t size[P_COUNT];
  size_t bucket0_end;      // = off[P_W3]
  int H, A;
  void init(int H_, int A_);
  size_t total() const { return off[P_COUNT]; }
  size_t reference_count() const; // libtorch parameters() element count
};
// reference (libtorch parameters() order, NCHW) <-> internal (NHWC-k order) permutation, on the host
void params_to_internal(const ParamLayout &L, const float *ref, float *internal);
void params_to_reference(const ParamLayout &L, const float *internal, float *ref);

// sample n of a batch lives in obs slot  (n / TP) * s1 + (n % TP) * s0 + base   (units: u32 pixels)
struct SampleMap {
  int TP;
  long s1, s0, base;
  int n0; // first sample of this launch (minibatch offset)
};

struct ProfClass {
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
};

struct Hyper {
  float clip, c_v, c_e, max_norm;
};

// Kernel-selection switches of ONE context (aleppo_set_option; initial values from the environment, read once in
// aleppo_create).  The launch helpers read them through tuning(): a thread-local pointer that every stateful entry
// point aims at its context first - the switches are per context, not per process.
struct Tuning {
  bool patch_conv = true;     // sample-stationary bf16 conv kernels (false: generic gather-GEMMs)
  bool fc_pipe = true;        // pipelined LDS-DMA fc GEMMs at minibatch sizes > 256
  bool fused_fwd = true;      // the update's conv1 -> conv2 -> conv3 forward as one launch (conv_fwd_fused.hpp)
  int fused_bwd = 1;          // conv2 dgrad + conv2 wgrad + conv1 wgrad as one launch (conv_bwd_fused.hpp): 0 never, 1 at
                              // minibatches >= 2048 samples (below, its fixed costs outweigh the bytes: v1.yaml's 1280), 2 always
  int fused_act = 1;          // frame ingest fused in front of the acting convolutions: 0 never, 1 where faster, 2 always
};
const Tuning &tuning();
void set_tuning(const Tuning *t); // nullptr -> process defaults
Tuning tuning_from_env();

struct Ctx {
  aleppo_config cfg{};
  int E = 0, T = 0, A = 0, H = 0, prec = 0, world = 1, rank = 0;
  long N = 0;          // E*T samples per rollout on this rank
  long maxB = 0;       // activation capacity in samples
  long batch_n = 0;    // samples currently in the training arrays
  ParamLayout L;
  std::string err;
  hipStream_t stream = nullptr, comm_stream = nullptr, wg_stream = nullptr;
  hipEvent_t ev_bucket0 = nullptr, ev_comm0 = nullptr, ev_comm1 = nullptr, ev_tmp = nullptr;
  hipEvent_t ev_head = nullptr, ev_dz3 = nullptr, ev_dz2 = nullptr, ev_wg = nullptr; // dgrad chain -> wgrad side stream
  hipEvent_t ev_adam = nullptr, ev_pack = nullptr; // adam done -> dgrad weight repack on the side stream -> done
  void *nccl_comm = nullptr;

  // ---- rollout storage (time-major scalars, env-major packed observation slots) ----
  uint32_t *obs = nullptr;     // [E][T+1][7056] packed stacks; slot T = bootstrap observation
  uint8_t *step_rec = nullptr; // [T] records of { float r[E]; u8 term[E]; u8 trunc[E]; u8 start[E] }
  size_t step_rec_bytes = 0;
  // float planes of the rollout buffer are stored as RT = float or IEEE half (cfg.rollout_precision): rt16 / rsz
  bool rt16 = false;
  size_t rsz = 4;              // bytes per stored plane element
  void *values_tm = nullptr;   // RT [T+1][E]
  void *logits_tm = nullptr;   // RT [T+1][E][A]
  int *actions_tm = nullptr;   // [T][E]
  uint8_t *lut = nullptr;      // [256]
  uint8_t *d_start = nullptr;  // [E] episode-start flags of the slot being ingested
  uint8_t *d_frames = nullptr; // staging for host frames (max(E*2*210*160))
  float *d_noise = nullptr;    // [2][E][A] (two staging halves: a slot may be enqueued while the previous one is pending)
  int *d_err = nullptr;        // device error word (flag overlap)
  unsigned int *d_done = nullptr; // arrival counter of the head kernel's ticket publish
  long long ticket = 0;        // last ticket published to h_actions[E]
  // pinned, GPU-visible hand-off words of the slot-ahead gate (gate_kernel): [0] release sequence (host writes),
  // [1] time-out report (the gate writes the sequence number it gave up on; 0 = none)
  unsigned long long *h_go = nullptr;
  unsigned long long go_seq = 0; // last sequence number a gate was enqueued for (64 bits: never wraps)
  unsigned long long gate_timeout_ticks = 0; // the gate's exit condition, in 100 MHz wall-clock ticks
  // a failure that leaves the rollout / learner state undefined (a gate that timed out, an update that failed half way):
  // every later stateful call but aleppo_destroy returns this error again
  bool failed = false;
  std::string fail_msg;
  unsigned noise_flip = 0;
  // aleppo_arm_step / aleppo_release_step: a step (and the next slot's acting kernels) enqueued behind the release word
  bool armed = false;
  const uint8_t *armed_start = nullptr; // the caller's mapped episode-start bytes of the armed step
  int act_queued_slot = -1;             // slot whose acting kernels + head are already enqueued (aleppo_act only waits)
  long long act_queued_ticket = 0;
  // pinned host staging
  int64_t *h_actions = nullptr; // [E]
  uint8_t *h_step = nullptr;    // one step record (aleppo_record_step path)
  uint8_t *h_rec = nullptr;     // [T] step records filled by aleppo_step, uploaded once at finish_rollout
  int rec_uploaded = 0;         // slots of h_rec already in device memory
  int rec_direct = 0;           // slots written straight to the device by aleppo_record_step
  uint8_t *h_frames = nullptr;
  float *h_noise = nullptr;
  int *h_err = nullptr;
  int pre_acted = -1;     // slot whose conv stack + split-K fc already ran (fused into aleppo_step); -1: none
  int t = 0;              // next slot to fill
  bool need_carry = false; // copy slot T -> slot 0 before the next rollout's first act
  uint64_t rng_counter = 0;
  // ---- training arrays, env-major n = e*T + t ----
  void *adv_n = nullptr, *ret_n = nullptr, *oldlp_n = nullptr; // RT [N], [N], [N][A]
  int *act_n = nullptr;
  uint8_t *mask_n = nullptr;
  float *mask_counts = nullptr; // [M_max] global unmasked count per minibatch
  // ---- network state ----
  float *P = nullptr, *G = nullptr, *Gs = nullptr, *M1 = nullptr, *M2 = nullptr; // fp32, internal layout (Gs: unused)
  void *Pc = nullptr;   // compute copy of P in T (same layout); == P for fp32
  void *W2d = nullptr, *W3d = nullptr, *WfcT = nullptr; // dgrad-transposed copies in T
  int64_t adam_step = 0;
  // ---- activations (T unless noted) ----
  void *a1 = nullptr, *a2 = nullptr, *a3 = nullptr; // [maxB][400][32], [maxB][81][64], [maxB][49][64]
  float *h = nullptr;                               // [maxB][H] fp32
  float *hpart = nullptr;                           // [FC_SPLITS][E][H] acting-size fc partial sums
  void *dh = nullptr;                               // [maxB][H] T
  void *dz3 = nullptr, *dz2 = nullptr, *dz1 = nullptr;
  float *logits_b = nullptr, *values_b = nullptr;   // [maxB][A], [maxB] (forward-only API / debug)
  // ---- gradient slabs ----
  float *slab = nullptr;
  size_t slab_floats = 0;
  size_t slab_off[11] = {0}; // W1 b1 W2 b2 W3 b3 Wfc bfc Wh bh
  float *adv_stats = nullptr;
  float *sumsq_part = nullptr; // [1024]
  // ---- per-sample train metrics [mi][B] x 5, and reduced [mi][8] ----
  float *metric_ps = nullptr;
  size_t metric_cap = 0; // floats per field
  size_t metric_red_cap = 0;
  float *metric_red = nullptr, *h_metric_red = nullptr;
  float *grad_norms = nullptr; // [mi]
  float *adam_sched = nullptr, *h_adam_sched = nullptr; // [mi][2] step scalars of one aleppo_train call (device / pinned)
  // ---- captured update (ALEPPO_OPT_UPDATE_GRAPH): the epochs x minibatches loop as one hipGraph, re-captured when
  // the shape (or a baked pointer) changes; the first call of a shape runs eagerly (one-time kernel attribute set-up)
  bool update_graph = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  struct GraphKey {
    int epochs = 0, M = 0, two = 0;
    long N = 0;
    const void *metric_ps = nullptr, *metric_red = nullptr;
    bool operator==(const GraphKey &o) const {
      return epochs == o.epochs && M == o.M && two == o.two && N == o.N && metric_ps == o.metric_ps &&
             metric_red == o.metric_red;
    }
  } graph_key, warm_key;
  long graph_replays = 0;
  int last_epochs = 0, last_M = 0;
  long last_B = 0;
  // ---- profiling ----
  Tuning tune;
  hipError_t async_err = hipSuccess; // first failure of a call whose status could not be returned on the spot
  // ---- boundary staging (aleppo_set_batch / aleppo_forward / aleppo_read_batch): grown on demand and never freed before
  // aleppo_destroy - hipFree / hipHostFree wait for EVERY stream of the device, including another context's stream
  // parked behind its release word, so no entry point but destroy calls them
  std::vector<void *> retired, retired_host;
  void *rb_tmp[2] = {nullptr, nullptr}; // aleppo_read_batch scratch
  size_t rb_cap[2] = {0, 0};
  uint8_t *stage_u8 = nullptr;   // NCHW uint8 observations as uploaded
  size_t stage_u8_cap = 0;
  uint32_t *stage_obs = nullptr; // packed stacks of aleppo_forward's samples (never the rollout slots)
  size_t stage_obs_cap = 0;
  bool prof_on = false;
  bool serial_update = false; // ALEPPO_OPT_SERIAL_UPDATE: every update kernel on the main stream
  bool dbg_no_publish = false;
  bool force_comm = false;
  ProfClass prof[ALEPPO_K_COUNT];
};

int set_err(Ctx *c, int code, const std::string &msg);
#define HIPCHK(c, x)                                                                                                   \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess)                                                                                              \
      return set_err((c), ALEPPO_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                           \
  } while (0)

// ------------------------------------------------------------------ kernel launchers (kernels.hip)
constexpr int MAX_ENVS_PER_RANK = 8192;
struct StartBits { // episode-start flags of one slot as a kernel argument (bit e of word e/32)
  uint32_t w[MAX_ENVS_PER_RANK / 32];
};
void launch_ingest(hipStream_t s, bool raw, const uint8_t *frames, const uint8_t *lut, const uint8_t *start,
                   const StartBits *sbits, uint32_t *obs, int E, int slots, int t_src, int t_dst);
void launch_copy_slot(hipStream_t s, uint32_t *obs, int E, int slots, int src, int dst);
void launch_gate(hipStream_t s, unsigned long long *go_dev, unsigned long long seq, unsigned long long timeout_ticks);
void launch_infer_head(hipStream_t s, const float *hpart, int nsplit, const float *bfc, const float *Wh,
                       const float *bh, const float *noise, uint64_t seed, uint64_t counter, void *logits_t,
                       void *values_t, int *actions_t, int64_t *pinned, unsigned int *done_ctr, long long ticket, int E,
                       int H, int A, const float *probs_in = nullptr, bool rt16 = false);
void launch_gae(hipStream_t s, uint8_t *step_rec, size_t rec_bytes, const void *values_tm, const void *logits_tm,
                const int *actions_tm, void *adv_n, void *ret_n, void *oldlp_n, int *act_n, uint8_t *mask_n, int *err,
                int E, int T, int A, float gamma, float lambda, bool clamp = true, bool rt16 = false);

void launch_adv_norm(hipStream_t s, void *adv_n, const uint8_t *mask_n, float *stats, long n, int phase, bool rt16);
void launch_plane_to_float(hipStream_t s, const void *src, float *dst, long n, bool rt16);
void launch_plane_from_float(hipStream_t s, const float *src, void *dst, long n, bool rt16);
void launch_mask_count(hipStream_t s, const uint8_t *mask_n, float *counts, long B, int M);
void launch_head_train(hipStream_t s, const float *h, const float *Wh, const float *bh, const int *act,
                       const void *oldlp, const void *adv, const void *ret, const uint8_t *mask,
                       const float *mask_count, Hyper hp, void *dh, int prec, float *ps_total, float *ps_clipped,
                       float *ps_value, float *ps_entropy, float *ps_ratio, float *slab_w, float *slab_b, int nblk,
                       long B, int H, int A, float *logits_out, float *values_out, int hparts = 1,
                       bool rt16 = false); // oldlp / adv / ret are f16 planes
struct ReduceSeg {
  const float *slab;
  int S;
  long n;
  long dst;
};
void launch_reduce_slabs(hipStream_t s, const ReduceSeg *segs, int nseg, float *G);
// squares G[0, n_main) in nblk_main blocks + the two tail tensors (slab sums fused when tail[i].slab != nullptr);
// returns the number of partials written (<= 1024)
int launch_sumsq(hipStream_t s, float *G, long n_main, float *partials, int nblk_main, const ReduceSeg tail[2],
                 int which = 0);
// clip + Adam over the whole flat vector; also refreshes the compute copy Pc (bf16) and the dgrad-side transposed weight
// layouts WfcT / W3d / W2d (both precisions) from the updated parameters in the same pass
void launch_adam(hipStream_t s, float *P, const float *G_in, float *G_out_scaled, float *M1, float *M2, void *Pc,
                 void *WfcT, void *W3d, void *W2d, const ParamLayout &L, int prec, const float *partials, int nblk,
                 float max_norm,
                 const float *sched, // device: { lr / (1 - beta1^t), sqrt(1 - beta2^t) } of this step
                 float beta1, float beta2, float eps, float *grad_norm_out);
void launch_pack_dgrad(hipStream_t s, const float *P, const ParamLayout &L, void *W2d, void *W3d, void *WfcT,
                       int prec);
void launch_cast_params(hipStream_t s, const float *P, void *Pc, long n);
void launch_metrics_reduce(hipStream_t s, const float *ps, size_t field_stride, const uint8_t *mask_n, long B, int M,
                           int epochs, float *out);
void launch_obs_unpack(hipStream_t s, const uint32_t *obs, uint8_t *out, long nsamp, SampleMap map);
void launch_obs_pack(hipStream_t s, const uint8_t *in, uint32_t *obs, long nsamp, SampleMap map);
void launch_transpose_tm_pitched(hipStream_t s, const void *src_tm, size_t pitch, void *dst_em, int E, int T, int inner,
                                 int elem);
void launch_heads_fwd(hipStream_t s, const float *h, const float *Wh, const float *bh, float *logits, float *values,
                      long n, int H, int A);
void launch_logsoftmax_rows(hipStream_t s, const float *in, float *out, long rows, int A);
// float-in / float-out free functions of ai::vision (library-only in the reference, vision.cc:8-32,:71-84)
void launch_area_resize(hipStream_t s, const float *in, float *out, long n);
void launch_rgb_to_gray(hipStream_t s, const float *in, float *out, long n);

// ------------------------------------------------------------------ GEMM launchers (gemm_launch.hip)
// prec selects T (ALEPPO_FP32 / ALEPPO_BF16); all pointers are device pointers of the matching type.
void conv1_fwd(hipStream_t s, int prec, const uint32_t *obs, SampleMap map, const void *W1, const float *b1, void *a1,
               long ns);
void conv2_fwd(hipStream_t s, int prec, const void *a1, const void *W2, const float *b2, void *a2, long ns);
void conv3_fwd(hipStream_t s, int prec, const void *a2, const void *W3, const float *b3, void *a3, long ns);
// returns the number of split-K partial slabs written to h ([parts][ns][H], slab 0 carries the bias); 1 unless
// max_parts > 1 allows the pipelined bf16 kernel to split K (the consumer then adds the slabs)
int fc_fwd(hipStream_t s, int prec, const void *a3, const void *Wfc, const float *bfc, float *h, long ns, int H,
           int max_parts = 1);
constexpr int FC_FWD_MAX_PARTS = 4;
void fc_fwd_splitk(hipStream_t s, int prec, const void *a3, const void *Wfc, float *hpart, long ns, int H);
void fc_dgrad(hipStream_t s, int prec, const void *dh, const void *WfcT, const void *a3, void *dz3, long ns, int H);
void conv3_dgrad(hipStream_t s, int prec, const void *dz3, const void *W3d, const void *a2, void *dz2, long ns);
void conv2_dgrad(hipStream_t s, int prec, const void *dz2, const void *W2d, const void *a1, void *dz1, long ns);
// wgrads write split-K slabs; return the number of slices S used (slab holds S*[M*N] then bias S*[M])
int fc_wgrad(hipStream_t s, int prec, const void *dh, const void *a3, float *slab_w, float *slab_b, long ns, int H);
int fc_wgrad_slices(int prec, long ns); // number of split-K slices fc_wgrad will use for ns samples
int conv3_wgrad(hipStream_t s, int prec, const void *dz3, const void *a2, float *slab_w, float *slab_b, long ns);
int conv2_wgrad(hipStream_t s, int prec, const void *dz2, const void *a1, float *slab_w, float *slab_b, long ns);
int conv1_wgrad(hipStream_t s, int prec, const void *dz1, const uint32_t *obs, SampleMap map, float *slab_w,
                float *slab_b, long ns);

// ------------------------------------------------------------------ sample-stationary bf16 conv kernels (conv_patch_launch.hip)
inline bool use_patch_kernels() { return tuning().patch_conv; } // false: ALEPPO_OPT_GENERIC_CONV / ALEPPO_GENERIC_CONV=1
void patch_conv1_fwd(hipStream_t s, const uint32_t *obs, SampleMap map, const void *W1, const float *b1, void *a1,
                     long ns);
void patch_conv2_fwd(hipStream_t s, const void *a1, const void *W2, const float *b2, void *a2, long ns);
void patch_conv3_fwd(hipStream_t s, const void *a2, const void *W3, const float *b3, void *a3, long ns);
void patch_conv3_dgrad(hipStream_t s, const void *dz3, const void *W3d, const void *a2, void *dz2, long ns);
void patch_conv2_dgrad(hipStream_t s, const void *dz2, const void *W2d, const void *a1, void *dz1, long ns);
// conv1 -> conv2 -> conv3 of the acting batch in one launch; ingest_mode 1 / 2 first forms every environment's stack from
// a new 84x84 frame / raw frame pair (fused frame ingest: see ActIngestParams in conv_patch.hpp)
void patch_fwd_fused(hipStream_t s, const uint32_t *obs, SampleMap map, const void *W1, const float *b1, const void *W2,
                     const float *b2, const void *W3, const float *b3, void *a1, void *a2, void *a3, long ns);
int patch_conv_bwd_fused(hipStream_t s, const void *dz2, const void *a1, const uint32_t *obs, SampleMap map, const void *W2d,
                         float *sw2, float *sb2, float *sw1, float *sb1, long ns);
void patch_act_convs(hipStream_t s, uint32_t *obs, SampleMap map, const void *W1, const float *b1, const void *W2,
                     const float *b2, const void *W3, const float *b3, void *a3, long ns, int ingest_mode = 0,
                     const uint8_t *frames = nullptr, const uint8_t *lut = nullptr, const StartBits *sbits = nullptr,
                     long src_delta = 0, const uint8_t *start_bytes = nullptr);
int patch_conv1_wgrad(hipStream_t s, const void *dz1, const uint32_t *obs, SampleMap map, float *sw, float *sb,
                      long ns);
int patch_conv2_wgrad(hipStream_t s, const void *dz2, const void *a1, float *sw, float *sb, long ns);
int patch_conv3_wgrad(hipStream_t s, const void *dz3, const void *a2, float *sw, float *sb, long ns);

} // namespace aleppo

struct aleppo_ctx : aleppo::Ctx {};
