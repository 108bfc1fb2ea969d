/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.
 * CPU restatement (plain C) of the reference's hot-path algorithm.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (ale-libtorch-ppo_amd/, include/aleppo.h) never links or calls it.
 *
 * Parity status: PINNED.  Checked by tests/test_oracle.py against (i) the six known-answer
 * cases of the reference's test/ai/gae-test.cc, (ii) the constant-preservation cases of
 * test/ai/vision-test.cc and (iii) tests/golden/ref_golden.npz = outputs of the reference's own
 * compiled gae.cc / buffer.cc / ppo/losses.cc / ppo/train.{h,cc} (oracle/_ref, built from
 * /root/reference by oracle/build_ref.sh; fixtures packed by oracle/make_golden.py).
 *
 * All tensors use the REFERENCE's layouts: env-major [E,T], observations NCHW uint8
 * [N,4,84,84], parameters flat in libtorch parameters() order
 *   conv1.w[32,4,8,8] conv1.b[32] conv2.w[64,32,4,4] conv2.b[64] conv3.w[64,64,3,3] conv3.b[64]
 *   fc.w[H,3136] fc.b[H] action.w[A,H] action.b[A] value.w[1,H] value.b[1]
 */
#ifndef ALEPPO_ORACLE_H
#define ALEPPO_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

size_t oracle_param_count(int H, int A);
/* offsets[13]: start of each of the 12 tensors + total */
void oracle_param_offsets(int H, int A, size_t *offsets);

/* src/ai/gae.cc:4-80.  returns 0, or -1 if start/terminal/truncation overlap (the reference throws) */
int oracle_gae(float *adv, const float *rewards, const float *values, const float *next_values,
               const uint8_t *terminals, const uint8_t *truncations, const uint8_t *episode_starts, int E, int T,
               float gamma, float lambda);
/* src/ai/buffer.cc:58-77: clamp rewards in place, gae, returns = adv + values, masks = !starts */
int oracle_buffer_get(float *rewards, const float *values, const float *next_values, const uint8_t *terminals,
                      const uint8_t *truncations, const uint8_t *episode_starts, float *adv, float *returns,
                      uint8_t *masks, int E, int T, float gamma, float lambda);
/* src/ai/rollout.cc:184-196 on obs [E,S,84,84] */
void oracle_update_observations(uint8_t *obs, const uint8_t *frames, const uint8_t *start, int E, int S);
/* src/ai/vision.cc:8-32: interpolate(mode=area) == adaptive average pooling */
void oracle_area_resize_f32(const float *in, float *out, int N, int IH, int IW, int OH, int OW);
/* src/ai/vision.cc:51,71-84: [N,3,P] -> [N,P] */
void oracle_rgb_to_gray(const float *rgb, float *out, int N, int P);
/* device preprocessing spec: per frame lut -> area 210x160->84x84 -> rint -> u8,
 * then max over the nframes frames (src/ai/environment/max_and_skip.cc:33-42). lut may be NULL. */
void oracle_preprocess_u8(const uint8_t *raw, const uint8_t *lut, uint8_t *out, int E, int nframes);

/* src/ai/ppo/losses.cc:45-47 */
void oracle_log_softmax(const float *logits, float *out, int B, int A);
void oracle_softmax(const float *logits, float *out, int B, int A);
/* src/bin/train.cc:374-375: multinomial(probs,1,true) == argmax(p/q), q~Exp(1), first max wins */
void oracle_sample(const float *probs, const float *q, int64_t *actions, int E, int A);

/* src/bin/train.cc:255-265.  acts (optional) receives the saved activations used by backward:
 * per sample x0[4*84*84] a1[32*400] a2[64*81] a3[3136] h[H] */
size_t oracle_acts_per_sample(int H);
void oracle_net_forward(const float *params, int H, int A, const uint8_t *obs, int N, float *logits, float *values,
                        float *acts);
/* gradients of sum_i (dlogits_i . logits_i + dvalues_i * value_i) wrt params, accumulated into grads (zeroed first) */
void oracle_net_backward(const float *params, int H, int A, int N, const float *acts, const float *dlogits,
                         const float *dvalues, float *grads);

/* src/ai/ppo/losses.cc:4-43 forward, closed-form backward.  n_mask <= 0: use sum(masks).
 * per-sample outputs may be NULL.  returns the scalar loss. */
float oracle_ppo_loss(const float *logits, const float *old_logp, const int64_t *actions, const float *adv,
                      const float *values, const float *returns, const uint8_t *masks, int B, int A, float clip,
                      float c_v, float c_e, float n_mask, float *clipped, float *value_losses, float *entropies,
                      float *total_losses, float *ratio, float *dlogits, float *dvalues);

/* src/ai/ppo/train.cc:12-46: returns the pre-clip total norm, scales grads in place */
float oracle_clip_grad_norm(float *grads, int H, int A, float max_norm);
/* torch::optim::Adam step (src/bin/train.cc:360-362), step counts from 1 */
void oracle_adam_step(float *p, const float *g, float *m, float *v, size_t n, double lr, double beta1, double beta2,
                      double eps, int64_t step);

/* src/ai/ppo/train.h:114-157: epochs x M contiguous minibatches.  out arrays sized
 * loss[epochs*M], grad_norm[epochs*M], per-sample [epochs*M*B] (may be NULL). */
int oracle_train(float *params, float *adam_m, float *adam_v, int64_t *adam_step, int H, int A, const uint8_t *obs,
                 const int64_t *actions, const float *old_logp, const float *adv, const float *returns,
                 const uint8_t *masks, int N, int epochs, int M, double lr, float clip, float c_v, float c_e,
                 float max_norm, float *loss, float *grad_norm, float *total_losses, float *ratio, float *entropies,
                 float *value_losses, float *clipped, float *last_grads);
/* advantage normalisation over unmasked samples - an extension with NO reference counterpart:
 * PARITY UNPINNED, see oracle.c.  returns the unmasked count. */
long oracle_adv_norm(float *adv, const uint8_t *masks, long n);
int oracle_num_threads(void);
#ifdef __cplusplus
}
#endif
#endif
