/* aleppo.h - C ABI of the MI355X-native PPO-over-ALE hot path (libaleppo.so).
 *
 * Drop-in boundary for the ONE data-parallel path of cemlyn007/ale-libtorch-ppo: vectorised
 * rollout pipeline (preprocess, 4-frame stack, rollout buffer, reward clamp + GAE + returns) and
 * the Nature-CNN actor-critic forward / backward with the PPO loss, global-norm clip and Adam.
 * The reference has no FFI; it reaches this path through ordinary C++ calls from main()
 * (src/bin/train.cc:320-465).  Each entry point below names the reference interface it replaces.
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: C linkage, opaque context, plain pointers and sizes, no C++ / torch types.
 * Every call returns ALEPPO_OK (0) or a negative aleppo_status; aleppo_last_error() gives the
 * message (the reference throws std::invalid_argument / std::runtime_error at the same places).
 * One owner thread per context; several contexts of one process may be driven from different threads at the same time.
 * After aleppo_create a context only ever touches its own HIP streams: no entry point but aleppo_destroy (and
 * aleppo_host_free) calls hipFree / hipHostFree / hipDeviceSynchronize or uses the null stream - those wait for every
 * stream of the device, another context's stream parked behind its release word included (aleppo_arm_step).  The same
 * rule binds the caller: while a step is armed, its owner thread must be the one that releases it, and no OTHER call of
 * that thread may wait for the device.  Host pointers are caller-owned and may be reused as soon as the
 * call returns, unless stated.  Tensors crossing the boundary use the REFERENCE's layouts
 * (env-major [E,T,...], NCHW uint8 observations, libtorch parameters() order); internal HBM
 * layouts are private.
 */
#ifndef ALEPPO_H
#define ALEPPO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ALEPPO_ABI_VERSION 2

typedef enum {
  ALEPPO_OK = 0,
  ALEPPO_ERR_INVALID_ARGUMENT = -1, /* std::invalid_argument in the reference */
  ALEPPO_ERR_RUNTIME = -2,          /* std::runtime_error in the reference */
  ALEPPO_ERR_HIP = -3,              /* a HIP / RCCL runtime call failed */
  ALEPPO_ERR_NO_DEVICE = -4         /* no usable gfx950 device: there is NO CPU fallback */
} aleppo_status;

typedef enum { ALEPPO_FP32 = 0, ALEPPO_BF16 = 1 } aleppo_precision;
/* Storage type of the rollout buffer's float planes (values, logits, advantages, returns, old log-probs:
 * Buffer's f32 tensors, src/ai/buffer.cc:12-38).  FP16 = BASELINE configs[4] "fp16 rollout buffer": the planes are
 * rounded to IEEE half when stored; reward clamp / GAE / returns / log-softmax arithmetic stays fp32 on the rounded
 * inputs; aleppo_read_batch still returns float.  The reference has only FP32. */
typedef enum { ALEPPO_ROLLOUT_FP32 = 0, ALEPPO_ROLLOUT_FP16 = 1 } aleppo_rollout_precision;
/* ALEPPO_HOST: ordinary or page-locked host memory, staged and copied by the call.  ALEPPO_DEVICE: device memory.
 * ALEPPO_HOST_MAPPED: page-locked host memory the GPU can address (hipHostMalloc / hipHostRegister'ed, e.g. the ring
 * the emulator threads write their frames into, src/ai/rollout.cc:325-326): the ingest kernel reads it in place over
 * the bus - no staging copy, no separate copy command on the slot's critical path. */
typedef enum { ALEPPO_HOST = 0, ALEPPO_DEVICE = 1, ALEPPO_HOST_MAPPED = 2 } aleppo_location;

/* What aleppo_push_frames receives per environment step. */
typedef enum {
  ALEPPO_FRAMES_84 = 0,      /* uint8 [E,84,84]: already gray + resized + max-pooled on the env threads
                                (what Rollout::step memcpy's into screen_buffers_, src/ai/rollout.cc:325-326) */
  ALEPPO_FRAMES_RAW_PAIR = 1 /* uint8 [E,2,210,160]: the last two emulator frames of the skip window as ALE
                                palette / gray bytes; the device applies the 256-entry LUT, the 84x84 area
                                resize and the 2-frame max (environment.cc:48-55, vision.cc:8-32,
                                max_and_skip.cc:33-42) */
} aleppo_frame_kind;

typedef struct aleppo_ctx aleppo_ctx;

/* Mirrors the hot-path part of Config (src/bin/train.cc:33-63) + Rollout ctor arguments
 * (src/ai/rollout.h:39-49).  New fields default (0) to reference behaviour. */
typedef struct {
  int32_t abi_version;    /* ALEPPO_ABI_VERSION */
  int32_t device_ordinal; /* HIP device; one process per GPU */
  int32_t world_size;     /* data-parallel ranks (1 = reference behaviour) */
  int32_t rank;
  int32_t num_envs;       /* E_local: environments owned by THIS rank (total_environments / world_size) */
  int32_t horizon;        /* T */
  int32_t num_actions;    /* A (the reference hard-codes 4, train.cc:36) */
  int32_t hidden_size;    /* H */
  int32_t frame_stack;    /* must be 4 (conv1 in-channels are hard-coded, train.cc:233) */
  int32_t precision;      /* aleppo_precision of the conv/linear stack */
  int32_t advantage_norm; /* 0 = none (the reference has none); 1 = normalise over masked samples */
  int32_t max_minibatch;  /* largest minibatch (samples per rank) aleppo_train will be asked for; 0 = E*T */
  int32_t rollout_precision; /* aleppo_rollout_precision; 0 = float planes like the reference */
  float gamma, lambda;    /* gae_discount, gae_lambda */
  float clip_param, value_loss_coef, entropy_coef, max_gradient_norm;
  float adam_beta1, adam_beta2, adam_eps; /* 0 -> 0.9 / 0.999 / 1e-5 (train.cc:360-362) */
  uint64_t seed;          /* counter-based sampling RNG when no external noise is supplied */
} aleppo_config;

/* Per-minibatch scalars = ai::ppo::train::Metrics reduced the way log_data() reduces them
 * (src/bin/train.cc:163-210): loss is the masked-mean loss tensor, the others masked means. */
typedef struct {
  float loss;          /* Metrics.loss[e][m] */
  float grad_norm;     /* Metrics.clipped_gradients[e][m] = PRE-clip total norm (train.cc:32-45) */
  float clipped_loss;  /* masked mean of clipped surrogate objective */
  float value_loss;    /* masked mean of 0.5 (v-R)^2 */
  float entropy;       /* masked mean entropy */
  float ratio;         /* masked mean probability ratio */
  float mask_count;    /* number of unmasked samples (global over ranks) */
} aleppo_minibatch_metrics;

/* Fields of the rollout batch (ai::buffer::Batch, src/ai/buffer.h:5-25, after prepare_batch,
 * src/bin/train.cc:272-283) readable with aleppo_read_batch, all in reference layout. */
typedef enum {
  ALEPPO_F_OBSERVATIONS = 0, /* uint8  [E,T,4,84,84] */
  ALEPPO_F_ACTIONS = 1,      /* int64  [E,T] */
  ALEPPO_F_REWARDS = 2,      /* float  [E,T] (clamped after finish_rollout) */
  ALEPPO_F_MASKS = 3,        /* uint8  [E,T] = !episode_starts */
  ALEPPO_F_LOGITS = 4,       /* float  [E,T,A] */
  ALEPPO_F_VALUES = 5,       /* float  [E,T] */
  ALEPPO_F_ADVANTAGES = 6,   /* float  [E,T] */
  ALEPPO_F_RETURNS = 7,      /* float  [E,T] */
  ALEPPO_F_LOG_PROBS = 8,    /* float  [E,T,A] = normalize_logits(logits) */
  ALEPPO_F_TERMINALS = 9,    /* uint8  [E,T] */
  ALEPPO_F_TRUNCATIONS = 10, /* uint8  [E,T] */
  ALEPPO_F_CURRENT_OBS = 11, /* uint8  [E,4,84,84]: Rollout::observations_ right now */
  ALEPPO_F_NEXT_VALUES = 12  /* float  [E]: bootstrap values of the last finish_rollout */
} aleppo_field;

/* Per-sample training metrics (ai::ppo::train::Metrics, src/ai/ppo/train.h:64-109), [epochs,M,B] */
typedef enum {
  ALEPPO_M_TOTAL_LOSSES = 0,
  ALEPPO_M_CLIPPED_LOSSES = 1,
  ALEPPO_M_VALUE_LOSSES = 2,
  ALEPPO_M_ENTROPIES = 3,
  ALEPPO_M_RATIO = 4
} aleppo_metric_field;

/* ------------------------------------------------------------------ lifetime */
int aleppo_abi_version(void);
/* ALEPPO_OK if HIP device `device_ordinal` exists and is a gfx950; ALEPPO_ERR_NO_DEVICE / _INVALID_ARGUMENT otherwise
 * (the check aleppo_create makes, on its own: what main() learns from torch::cuda::is_available(), train.cc:336-345). */
int aleppo_device_check(int device_ordinal);
/* Replaces the construction done in main(): Network + Adam + Rollout(+Buffer) (train.cc:358-387). */
int aleppo_create(const aleppo_config *cfg, aleppo_ctx **out);
void aleppo_destroy(aleppo_ctx *ctx);
/* Message of the last failed call on ctx (ctx may be NULL for aleppo_create failures). */
const char *aleppo_last_error(const aleppo_ctx *ctx);

/* ------------------------------------------------------------------ parameters
 * Flat float32 in libtorch parameters() order of NetworkImpl (train.cc:230-253):
 * sequential.{0,2,4,7}.{weight,bias}, action_head.{weight,bias}, value_head.{weight,bias}. */
int aleppo_param_count(const aleppo_ctx *ctx, size_t *count);
int aleppo_load_params(aleppo_ctx *ctx, const float *flat, size_t count);   /* also resets Adam state */
int aleppo_export_params(aleppo_ctx *ctx, float *flat, size_t count);
/* Gradient of the LAST minibatch as clip_grad_norm_ left it (scaled), same order. Parity dumps. */
int aleppo_export_grads(aleppo_ctx *ctx, float *flat, size_t count);

/* Checkpoint / resume (the reference has none): Adam moments in the same order as the
 * parameters plus the step count; together with aleppo_export_params this is the whole learner state. */
int aleppo_export_optimizer(aleppo_ctx *ctx, float *exp_avg, float *exp_avg_sq, int64_t *step, size_t count);
int aleppo_import_optimizer(aleppo_ctx *ctx, const float *exp_avg, const float *exp_avg_sq, int64_t step,
                            size_t count);

/* ------------------------------------------------------------------ rollout (Rollout::rollout, rollout.cc:198-278)
 * Per slot t = 0..T-1 the caller does  act -> (step its emulators) -> push_frames -> record_step,
 * then finish_rollout.  aleppo_step = push_frames + record_step in one upload. */

/* Action selector (train.cc:367-379): eval forward on the current stack, softmax,
 * multinomial(1, replacement) = argmax(p/q).  noise: float [E,A] of Exp(1) draws (host) to reproduce a
 * captured stream bit-exactly, or NULL for the built-in counter-based generator.
 * *actions_pinned: int64 [E] in page-locked host memory owned by ctx, valid when the call returns and
 * until the next aleppo_act; env worker threads may read it concurrently (replaces the per-env
 * actions[i].item<int64_t>() of rollout.cc:312-313).  Stores logits/values for slot t. */
int aleppo_act(aleppo_ctx *ctx, const float *noise, const int64_t **actions_pinned);

/* Rollout::update_observations (rollout.cc:184-196) fused with Buffer::add's observation copy
 * (buffer.cc:47) and, for ALEPPO_FRAMES_RAW_PAIR, the preprocessing the reference does on env threads.
 * episode_start: uint8 [E] = is_episode_start_cpu_ at ENTRY of this slot (rollout.cc:190). */
int aleppo_push_frames(aleppo_ctx *ctx, const uint8_t *frames, int frame_kind, int location,
                       const uint8_t *episode_start);
/* Page-locked host memory the GPU can address, for the buffers the emulator worker threads write their frames into
 * (replaces Rollout::screen_buffers_, src/ai/rollout.cc:325-326): pass it to aleppo_step / aleppo_push_frames with
 * ALEPPO_HOST_MAPPED and the ingest kernel reads the frames in place - no staging copy on the slot's critical path.
 * The caller must not rewrite a buffer before the aleppo_act / aleppo_finish_rollout that follows its aleppo_step has
 * returned (the emulators are stepped after aleppo_act, so the natural loop satisfies this with ONE buffer). */
int aleppo_host_alloc(aleppo_ctx *ctx, size_t bytes, void **ptr);
int aleppo_host_free(aleppo_ctx *ctx, void *ptr);
/* 256-entry palette -> gray LUT used by ALEPPO_FRAMES_RAW_PAIR (default: identity). */
int aleppo_set_gray_lut(aleppo_ctx *ctx, const uint8_t *lut256);

/* The per-env scalar writes of rollout.cc:212-227 + Buffer::add (buffer.cc:48-54) for slot t, as ONE
 * upload; advances t.  All arrays are host [E]; rewards of envs in an episode-start slot are whatever the
 * caller kept (the reference keeps the stale value, rollout.cc:214). */
int aleppo_record_step(aleppo_ctx *ctx, const float *rewards, const uint8_t *terminated,
                       const uint8_t *truncated, const uint8_t *episode_start);
int aleppo_step(aleppo_ctx *ctx, const uint8_t *frames, int frame_kind, int location, const float *rewards,
                const uint8_t *terminated, const uint8_t *truncated, const uint8_t *episode_start);

/* The same step with the stream running ONE SLOT AHEAD of the emulator (replaces the launch latency between
 * rollout.cc:312-313, the actions reaching the workers, and rollout.cc:204-208, the next forward pass).
 * aleppo_arm_step - called right after aleppo_act returned slot t's actions, BEFORE the emulators are stepped - enqueues
 * the ingest of the frames the emulators are about to write into `frames` and of the episode-start flags they are about
 * to write into `episode_start_mapped` (uint8 [E]; both in mapped page-locked memory from aleppo_host_alloc), plus slot
 * t+1's acting kernels (noise_next: that slot's sampling noise or NULL), all behind a one-wave gate kernel that polls a
 * release word in mapped host memory.  aleppo_release_step - called when the emulators are done - releases the
 * stream and records slot t's scalars (the per-env writes of rollout.cc:212-227; episode starts are read from
 * episode_start_mapped); it advances t.  The next aleppo_act only waits for the actions (its noise argument is ignored:
 * that head is already on the stream with noise_next).  Between the two calls every
 * other stateful entry point fails with ALEPPO_ERR_RUNTIME.  Results are bit-identical to aleppo_act / aleppo_step.
 * The gate has an exit condition: if it is not released within ALEPPO_OPT_GATE_TIMEOUT_MS (default 120 000) it gives
 * up, the stream drains, and the release (or the next aleppo_act / aleppo_finish_rollout) fails the context: from then
 * on every call returns ALEPPO_ERR_RUNTIME with that message until aleppo_destroy; the frame buffers must stay
 * allocated until then.  A null argument to aleppo_release_step is reported while the step is still armed (repeat the
 * call). */
int aleppo_arm_step(aleppo_ctx *ctx, const uint8_t *frames, int frame_kind, const uint8_t *episode_start_mapped,
                    const float *noise_next);
int aleppo_release_step(aleppo_ctx *ctx, const float *rewards, const uint8_t *terminated, const uint8_t *truncated);

/* Rollout::rollout()'s slot loop (rollout.cc:198-278) over a PRE-RECORDED environment trace: for t in [0, T):
 * aleppo_act (built-in RNG) then aleppo_step with slot t of the trace.  frames: DEVICE memory, slot t at
 * frames + t * slot_stride_bytes (16-byte aligned); rewards [T][E] f32 and terminated / truncated / episode_start
 * [T][E] u8 are HOST arrays.  frame_location: ALEPPO_DEVICE or ALEPPO_HOST_MAPPED (the whole trace in mapped
 * page-locked host memory: what a ring filled by emulator threads looks like to the device).  noise: float
 * [T][E][A] Exp(1) draws (host) for a reproducible action stream, or NULL for the built-in generator.  The sampled
 * actions do not influence a recorded trace, so this entry point only serves replay / throughput measurement with
 * the whole host loop native (like the reference's C++ loop); a live emulator calls aleppo_act / aleppo_step itself.
 * Leaves the context ready for aleppo_finish_rollout. */
int aleppo_replay_rollout(aleppo_ctx *ctx, const uint8_t *frames, int frame_kind, int frame_location,
                          size_t slot_stride_bytes, const float *rewards, const uint8_t *terminated,
                          const uint8_t *truncated, const uint8_t *episode_start, const float *noise);

/* Tail of Rollout::rollout (rollout.cc:268-270) + Buffer::get (buffer.cc:58-77) + prepare_batch
 * (train.cc:272-283): bootstrap forward (draws and discards one sample like the reference), reward
 * clamp, GAE, returns, masks, old log-probs.  ALEPPO_ERR_RUNTIME if the buffer is not full
 * (buffer.cc:64-65); ALEPPO_ERR_INVALID_ARGUMENT if flags overlap (gae.cc:49-53). */
int aleppo_finish_rollout(aleppo_ctx *ctx, const float *noise);

/* ------------------------------------------------------------------ update (ai::ppo::train::train, train.h:133-157)
 * epochs x num_mini_batches contiguous env-major slices; per minibatch forward, loss, backward,
 * [RCCL all-reduce when world_size>1], clip_grad_norm_, Adam.  lr is this rollout's annealed rate
 * (train.cc:424-428).  out_metrics: [epochs*num_mini_batches], may be NULL.
 * ALEPPO_ERR_RUNTIME if E*T % num_mini_batches != 0 (train.h:140-143). */
int aleppo_train(aleppo_ctx *ctx, double lr, int epochs, int num_mini_batches,
                 aleppo_minibatch_metrics *out_metrics);
/* Per-sample metric tensors of the last aleppo_train, float [epochs,M,B]. */
int aleppo_read_train_metric(aleppo_ctx *ctx, int metric_field, float *dst, size_t count);

/* Same update on a caller-supplied batch (what ai::ppo::train::train takes: train.h:34-57, 133-137):
 * observations uint8 [N,4,84,84], actions int64 [N], old log-probs float [N,A], advantages, returns
 * float [N], masks uint8 [N] (host).  N <= E*T. Used by the parity tests. */
int aleppo_set_batch(aleppo_ctx *ctx, const uint8_t *observations, const int64_t *actions,
                     const float *log_probabilities, const float *advantages, const float *returns,
                     const uint8_t *masks, int64_t n);

int aleppo_read_batch(aleppo_ctx *ctx, int field, void *dst, size_t bytes);
/* Network forward only (NetworkImpl::forward, train.cc:255-265) on host observations uint8
 * [n,4,84,84] -> logits float [n,A], values float [n].  n <= max(E, max_minibatch). */
int aleppo_forward(aleppo_ctx *ctx, const uint8_t *observations, int64_t n, float *logits, float *values);

/* ------------------------------------------------------------------ multi-GPU (no reference counterpart)
 * One process per GPU.  Rank 0 creates the id, the launcher broadcasts its bytes, every rank calls
 * aleppo_comm_init.  Gradients (+ mask counts) are all-reduced with RCCL inside aleppo_train. */
#define ALEPPO_UNIQUE_ID_BYTES 128
int aleppo_comm_unique_id(uint8_t id[ALEPPO_UNIQUE_ID_BYTES]);
int aleppo_comm_init(aleppo_ctx *ctx, const uint8_t id[ALEPPO_UNIQUE_ID_BYTES]);

/* ------------------------------------------------------------------ stateless operators (host in / host out)
 * The reference's free functions, for parity tests that read like the reference's own tests.  Each one converts
 * its host tensors to the hot path's device layout and launches the SAME kernel the rollout / update launches
 * (gae_kernel, ingest_kernel, head_train_kernel, infer_head_kernel): there is no second implementation. */
/* ai::gae::gae (src/ai/gae.h:4-7): env-major [E,T]; same validation errors (gae.cc:8-53). */
int aleppo_gae(int device_ordinal, float *advantages, const float *rewards, const float *values,
               const float *next_values, const uint8_t *terminals, const uint8_t *truncations,
               const uint8_t *episode_starts, int64_t num_envs, int64_t num_steps, float gamma, float lambda);
/* ai::vision::resize_frame_stacked_grayscale_images (vision.cc:22-32): float [n,210,160] -> [n,84,84] */
int aleppo_vision_resize_area(int device_ordinal, const float *images, float *out, int64_t n);
/* ai::vision::rgb_to_grayscale_frame_stacked_images (vision.cc:71-84): float [n,3,84,84] -> [n,84,84] */
int aleppo_vision_rgb_to_gray(int device_ordinal, const float *images, float *out, int64_t n);
/* The fused device preprocessing on its own: uint8 [n,2,210,160] (+lut or NULL) -> uint8 [n,84,84] */
int aleppo_preprocess(int device_ordinal, const uint8_t *raw_pairs, const uint8_t *lut256, uint8_t *out, int64_t n);
/* Rollout::update_observations (rollout.cc:184-196) on host tensors: obs uint8 [E,4,84,84] in/out */
int aleppo_update_observations(int device_ordinal, uint8_t *observations, const uint8_t *frames,
                               const uint8_t *episode_start, int64_t num_envs);
/* ai::ppo::losses::compute (+ normalize_logits) forward and gradients (losses.cc:4-47):
 * logits float [B,A] RAW (un-normalised); outputs per-sample [B]; dlogits [B,A]; any output may be NULL. */
int aleppo_ppo_loss(int device_ordinal, const float *logits, const float *old_log_probabilities,
                    const int64_t *actions, const float *advantages, const float *values, const float *returns,
                    const uint8_t *masks, int64_t batch, int64_t num_actions, float clip_param,
                    float value_loss_coef, float entropy_coef, float *loss, float *clipped, float *value_losses,
                    float *entropies, float *total_losses, float *ratio, float *dlogits, float *dvalues);
/* multinomial(probs,1,true) given its exponential noise (train.cc:374-375): probs,q float [E,A] */
int aleppo_sample(int device_ordinal, const float *probs, const float *q, int64_t *actions, int64_t num_envs,
                  int64_t num_actions);

/* ------------------------------------------------------------------ measurement hooks (bench.py)
 * Average device time in ms of the named kernel class over the calls since the last reset, measured
 * with HIP events on the stream the kernels run on; *launches gets the number of timed launches. */
typedef enum {
  ALEPPO_K_INGEST = 0,      /* preprocess + frame stack + rollout-slot write */
  ALEPPO_K_GAE = 1,         /* reward clamp + GAE + returns + old log-probs */
  ALEPPO_K_HEAD = 2,        /* heads + PPO loss forward/backward */
  ALEPPO_K_ADAM = 3,        /* sum of squares + clip + Adam + dgrad weight repack */
  ALEPPO_K_CONV1_FWD = 4,
  ALEPPO_K_CONV2_FWD = 5,
  ALEPPO_K_CONV3_FWD = 6,
  ALEPPO_K_FC_FWD = 7,
  ALEPPO_K_FC_DGRAD = 8,
  ALEPPO_K_FC_WGRAD = 9,
  ALEPPO_K_CONV3_DGRAD = 10,
  ALEPPO_K_CONV3_WGRAD = 11,
  ALEPPO_K_CONV2_DGRAD = 12,
  ALEPPO_K_CONV2_WGRAD = 13,
  ALEPPO_K_CONV1_WGRAD = 14,
  ALEPPO_K_REDUCE = 15,     /* split-K slab reduction */
  ALEPPO_K_INFER_HEAD = 16, /* action head + sampling */
  ALEPPO_K_ACT_FUSED = 17,  /* frame ingest + conv1-3 of the acting batch in one launch (aleppo_step, bf16) */
  ALEPPO_K_CONV_FWD = 18,   /* conv1 -> conv2 -> conv3 of the update's forward pass in one launch (bf16) */
  ALEPPO_K_CONV_BWD = 19,   /* conv2 dgrad + conv2 wgrad + conv1 wgrad of the update's backward pass in one launch (bf16) */
  ALEPPO_K_COUNT = 20
} aleppo_kernel_class;
int aleppo_profile_enable(aleppo_ctx *ctx, int on);
int aleppo_profile_read(aleppo_ctx *ctx, int kernel_class, double *avg_ms, int64_t *launches);
int aleppo_profile_reset(aleppo_ctx *ctx);
/* Per-context tuning / A-B switches.  ALEPPO_OPT_GENERIC_CONV = 1: run the bf16 convolutions on the generic
 * gather-GEMM kernels instead of the sample-stationary ones (same math, used by the parity tests). */
typedef enum {
  ALEPPO_OPT_GENERIC_CONV = 0,
  ALEPPO_OPT_DEBUG_NO_PUBLISH = 1, /* diagnosis only: the head kernel skips the pinned-memory hand-off */
  ALEPPO_OPT_FORCE_COMM = 2,       /* tests: run the RCCL all-reduce path even with a 1-rank communicator */
  ALEPPO_OPT_SERIAL_UPDATE = 3,    /* measurement: run the weight-gradient kernels on the main stream too (isolated
                                      per-kernel timings; default 0 = co-scheduled on a second stream) */
  ALEPPO_OPT_FC_PIPE = 4,          /* 0: small-tile fc GEMMs instead of the pipelined LDS-DMA ones (A/B, parity tests) */
  /* 5 and 8 were the opt-in pipelined fc weight gradient and the fused conv2-dgrad + conv1-wgrad launch: both measured
     slower than the defaults in rounds 1-3 and were deleted */
  ALEPPO_OPT_FUSED_ACT = 6,        /* frame ingest fused in front of the acting convolutions (bf16): 0 never, 1 where it
                                      is faster (default: given 84x84 frames, raw pairs in mapped host memory), 2 always */
  ALEPPO_OPT_GATE_TIMEOUT_MS = 9,  /* exit condition of the slot-ahead gate in milliseconds (default 120 000; also the
                                      environment variable ALEPPO_GATE_TIMEOUT_MS at aleppo_create) */
  ALEPPO_OPT_FUSED_FWD = 10,       /* 0: the update's forward convolutions as three launches instead of the fused
                                      conv1 -> conv2 -> conv3 kernel (bf16; same bits either way: A/B, parity tests; also
                                      the environment variable ALEPPO_FWD_FUSED at aleppo_create) */
  ALEPPO_OPT_FUSED_BWD = 11,       /* conv2's data gradient, conv2's weight gradient and conv1's weight gradient as ONE launch that
                                      keeps dz1 on the CU (bf16): 0 never (three launches on two streams: A/B, parity
                                      tests), 1 at minibatches of >= 2048 samples (default), 2 always; environment:
                                      ALEPPO_BWD_FUSED */
  ALEPPO_OPT_UPDATE_GRAPH = 7      /* 1: capture the epochs x minibatches loop of aleppo_train in a hipGraph and replay it
                                      (capture_train_cuda_graph, src/ai/ppo/train.h:163-195); lr and the Adam bias
                                      corrections are device scalars, so a replay follows the annealed rate */
} aleppo_option;
int aleppo_set_option(aleppo_ctx *ctx, int option, int value);
/* Current value of an option; for ALEPPO_OPT_UPDATE_GRAPH the number of graph launches so far (0 = every update ran
 * eagerly), for the others the value last set / the default. */
int aleppo_get_option(aleppo_ctx *ctx, int option, int64_t *value);
/* Block until everything enqueued on ctx's streams has finished. */
int aleppo_synchronize(aleppo_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
