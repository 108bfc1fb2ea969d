/* TEST INFRASTRUCTURE - NOT PRODUCT CODE.  See oracle.h for scope, pinning and layouts.
 * Plain-C restatement of the reference hot path; every function cites the reference
 * file:line it follows.  Built by oracle/Makefile into oracle/liboracle.so.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ parameters */
static const size_t CONV_SIZES[6] = {32 * 4 * 8 * 8, 32, 64 * 32 * 4 * 4, 64, 64 * 64 * 3 * 3, 64};

void oracle_param_offsets(int H, int A, size_t *o) {
  size_t s[12];
  for (int i = 0; i < 6; ++i)
    s[i] = CONV_SIZES[i];
  s[6] = (size_t)H * 3136;
  s[7] = (size_t)H;
  s[8] = (size_t)A * H;
  s[9] = (size_t)A;
  s[10] = (size_t)H;
  s[11] = 1;
  o[0] = 0;
  for (int i = 0; i < 12; ++i)
    o[i + 1] = o[i] + s[i];
}
size_t oracle_param_count(int H, int A) {
  size_t o[13];
  oracle_param_offsets(H, A, o);
  return o[12];
}

/* ------------------------------------------------------------------ GAE: src/ai/gae.cc:4-80 */
int oracle_gae(float *adv, const float *rewards, const float *values, const float *next_values,
               const uint8_t *terminals, const uint8_t *truncations, const uint8_t *episode_starts, int E, int T,
               float gamma, float lambda) {
  /* gae.cc:49-53: the three flags must be mutually exclusive */
  for (long i = 0; i < (long)E * T; ++i)
    if ((episode_starts[i] != 0) + (terminals[i] != 0) + (truncations[i] != 0) > 1)
      return -1;
  const float gl = gamma * lambda; /* gae.cc:62: float*float before touching the tensor */
  for (int e = 0; e < E; ++e) {
    float last = 0.0f;           /* gae.cc:57 */
    float nv = next_values[e];   /* gae.cc:58 */
    for (int i = T - 1; i >= 0; --i) {
      const long k = (long)e * T + i;
      const float r = rewards[k], v = values[k];
      const float boot = (r + gamma * nv) - v;      /* gae.cc:61-62 / 65-66 */
      float a = boot + gl * last;                   /* running, gae.cc:61-63 */
      if (episode_starts[k])
        a = 0.0f;                                   /* gae.cc:68-69 */
      if (terminals[k])
        a = r - v;                                  /* gae.cc:64, 70-71 */
      if (truncations[k])
        a = boot;                                   /* gae.cc:65-66, 72-73 */
      adv[k] = a;                                   /* gae.cc:75 */
      last = a;                                     /* gae.cc:77 */
      nv = v;                                       /* gae.cc:78 */
    }
  }
  return 0;
}

/* src/ai/buffer.cc:58-77 */
int oracle_buffer_get(float *rewards, const float *values, const float *next_values, const uint8_t *terminals,
                      const uint8_t *truncations, const uint8_t *episode_starts, float *adv, float *returns,
                      uint8_t *masks, int E, int T, float gamma, float lambda) {
  const long n = (long)E * T;
  for (long i = 0; i < n; ++i) /* buffer.cc:67 clamp_ in place */
    rewards[i] = rewards[i] < -1.0f ? -1.0f : (rewards[i] > 1.0f ? 1.0f : rewards[i]);
  int rc = oracle_gae(adv, rewards, values, next_values, terminals, truncations, episode_starts, E, T, gamma, lambda);
  if (rc)
    return rc;
  for (long i = 0; i < n; ++i) {
    returns[i] = adv[i] + values[i]; /* buffer.cc:70-71 */
    masks[i] = !episode_starts[i];   /* buffer.cc:74 */
  }
  return 0;
}

/* src/ai/rollout.cc:184-196 */
void oracle_update_observations(uint8_t *obs, const uint8_t *frames, const uint8_t *start, int E, int S) {
  const int P = 84 * 84;
  for (int e = 0; e < E; ++e) {
    uint8_t *o = obs + (size_t)e * S * P;
    const uint8_t *f = frames + (size_t)e * P;
    for (int k = S - 1; k > 0; --k) /* :185-188 shift towards older slots */
      memcpy(o + (size_t)k * P, o + (size_t)(k - 1) * P, P);
    if (start[e])                   /* :189-193 broadcast on an episode-start slot */
      for (int k = 0; k < S; ++k)
        memcpy(o + (size_t)k * P, f, P);
    memcpy(o, f, P);                /* :194-195 newest frame at index 0 */
  }
}

/* ------------------------------------------------------------------ vision */
/* src/ai/vision.cc:8-32 : interpolate(size, mode=area) = adaptive average pooling */
void oracle_area_resize_f32(const float *in, float *out, int N, int IH, int IW, int OH, int OW) {
  for (int n = 0; n < N; ++n)
    for (int i = 0; i < OH; ++i) {
      const int y0 = (i * IH) / OH, y1 = ((i + 1) * IH + OH - 1) / OH;
      for (int j = 0; j < OW; ++j) {
        const int x0 = (j * IW) / OW, x1 = ((j + 1) * IW + OW - 1) / OW;
        float s = 0.0f;
        for (int y = y0; y < y1; ++y)
          for (int x = x0; x < x1; ++x)
            s += in[((size_t)n * IH + y) * IW + x];
        out[((size_t)n * OH + i) * OW + j] = s / (float)((y1 - y0) * (x1 - x0));
      }
    }
}

/* src/ai/vision.cc:51, :71-84 */
void oracle_rgb_to_gray(const float *rgb, float *out, int N, int P) {
  const float w[3] = {0.2125f, 0.7154f, 0.0721f};
  for (int n = 0; n < N; ++n)
    for (int p = 0; p < P; ++p) {
      const float *s = rgb + (size_t)n * 3 * P + p;
      out[(size_t)n * P + p] = s[0] * w[0] + s[(size_t)P] * w[1] + s[(size_t)2 * P] * w[2];
    }
}

/* Device preprocessing spec: per emulator frame gray LUT (src/ai/environment/environment.cc:48-55) ->
 * area resize (vision.cc:8-32) -> round-half-even to u8; then the 2-frame max taken AFTER resizing
 * (src/ai/environment/max_and_skip.cc:33-42, Q12). */
void oracle_preprocess_u8(const uint8_t *raw, const uint8_t *lut, uint8_t *out, int E, int nframes) {
  const int IH = 210, IW = 160, OH = 84, OW = 84;
  for (int e = 0; e < E; ++e)
    for (int i = 0; i < OH; ++i) {
      const int y0 = (i * IH) / OH, y1 = ((i + 1) * IH + OH - 1) / OH;
      for (int j = 0; j < OW; ++j) {
        const int x0 = (j * IW) / OW, x1 = ((j + 1) * IW + OW - 1) / OW;
        int best = 0;
        for (int f = 0; f < nframes; ++f) {
          const uint8_t *src = raw + ((size_t)e * nframes + f) * IH * IW;
          float s = 0.0f;
          for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
              uint8_t g = src[y * IW + x];
              s += (float)(lut ? lut[g] : g);
            }
          int v = (int)rintf(s / (float)((y1 - y0) * (x1 - x0)));
          if (v > best)
            best = v;
        }
        out[((size_t)e * OH + i) * OW + j] = (uint8_t)(best > 255 ? 255 : best);
      }
    }
}

/* ------------------------------------------------------------------ softmax family */
/* src/ai/ppo/losses.cc:45-47 */
void oracle_log_softmax(const float *logits, float *out, int B, int A) {
  for (int i = 0; i < B; ++i) {
    const float *z = logits + (size_t)i * A;
    float mx = z[0];
    for (int k = 1; k < A; ++k)
      mx = z[k] > mx ? z[k] : mx;
    float s = 0.0f;
    for (int k = 0; k < A; ++k)
      s += expf(z[k] - mx);
    const float lse = mx + logf(s);
    for (int k = 0; k < A; ++k)
      out[(size_t)i * A + k] = z[k] - lse;
  }
}
void oracle_softmax(const float *logits, float *out, int B, int A) {
  for (int i = 0; i < B; ++i) {
    const float *z = logits + (size_t)i * A;
    float mx = z[0];
    for (int k = 1; k < A; ++k)
      mx = z[k] > mx ? z[k] : mx;
    float s = 0.0f;
    for (int k = 0; k < A; ++k)
      s += expf(z[k] - mx);
    for (int k = 0; k < A; ++k)
      out[(size_t)i * A + k] = expf(z[k] - mx) / s;
  }
}
/* src/bin/train.cc:374-375 */
void oracle_sample(const float *probs, const float *q, int64_t *actions, int E, int A) {
  for (int e = 0; e < E; ++e) {
    int best = 0;
    float bv = probs[(size_t)e * A] / q[(size_t)e * A];
    for (int k = 1; k < A; ++k) {
      const float v = probs[(size_t)e * A + k] / q[(size_t)e * A + k];
      if (v > bv) {
        bv = v;
        best = k;
      }
    }
    actions[e] = best;
  }
}

/* ------------------------------------------------------------------ network: src/bin/train.cc:230-265 */
#define X0 (4 * 84 * 84)
#define A1 (32 * 400)
#define A2 (64 * 81)
#define A3 3136
size_t oracle_acts_per_sample(int H) { return (size_t)X0 + A1 + A2 + A3 + (size_t)H; }

static void im2col(const float *in, int C, int IH, int IW, int K, int S, int OH, int OW, float *cols) {
  for (int c = 0; c < C; ++c)
    for (int kh = 0; kh < K; ++kh)
      for (int kw = 0; kw < K; ++kw) {
        float *row = cols + (size_t)((c * K + kh) * K + kw) * OH * OW;
        for (int oy = 0; oy < OH; ++oy)
          for (int ox = 0; ox < OW; ++ox)
            row[oy * OW + ox] = in[((size_t)c * IH + oy * S + kh) * IW + ox * S + kw];
      }
}
static void col2im_add(const float *cols, int C, int IH, int IW, int K, int S, int OH, int OW, float *in) {
  for (int c = 0; c < C; ++c)
    for (int kh = 0; kh < K; ++kh)
      for (int kw = 0; kw < K; ++kw) {
        const float *row = cols + (size_t)((c * K + kh) * K + kw) * OH * OW;
        for (int oy = 0; oy < OH; ++oy)
          for (int ox = 0; ox < OW; ++ox)
            in[((size_t)c * IH + oy * S + kh) * IW + ox * S + kw] += row[oy * OW + ox];
      }
}
/* out[oc][p] = relu(b[oc] + sum_k w[oc][k] cols[k][p]) */
static void conv_gemm_relu(const float *w, const float *b, const float *cols, int OC, int Kd, int P, float *out) {
  for (int oc = 0; oc < OC; ++oc) {
    float *o = out + (size_t)oc * P;
    for (int p = 0; p < P; ++p)
      o[p] = b[oc];
    for (int k = 0; k < Kd; ++k) {
      const float wk = w[(size_t)oc * Kd + k];
      const float *c = cols + (size_t)k * P;
      for (int p = 0; p < P; ++p)
        o[p] += wk * c[p];
    }
    for (int p = 0; p < P; ++p)
      o[p] = o[p] > 0.0f ? o[p] : 0.0f;
  }
}

static void forward_one(const float *prm, const size_t *po, int H, int A, const uint8_t *obs, float *logits,
                        float *value, float *acts, float *cols) {
  float *x0 = acts, *a1 = x0 + X0, *a2 = a1 + A1, *a3 = a2 + A2, *h = a3 + A3;
  for (int i = 0; i < X0; ++i)
    x0[i] = (float)obs[i] / 255.0f; /* train.cc:258-259 */
  im2col(x0, 4, 84, 84, 8, 4, 20, 20, cols);
  conv_gemm_relu(prm + po[0], prm + po[1], cols, 32, 256, 400, a1);
  im2col(a1, 32, 20, 20, 4, 2, 9, 9, cols);
  conv_gemm_relu(prm + po[2], prm + po[3], cols, 64, 512, 81, a2);
  im2col(a2, 64, 9, 9, 3, 1, 7, 7, cols);
  conv_gemm_relu(prm + po[4], prm + po[5], cols, 64, 576, 49, a3);
  for (int o = 0; o < H; ++o) { /* linear 3136->H, NO relu (Q3) */
    const float *w = prm + po[6] + (size_t)o * A3;
    float s = 0.0f;
    for (int j = 0; j < A3; ++j)
      s += w[j] * a3[j];
    h[o] = s + prm[po[7] + o];
  }
  for (int a = 0; a < A; ++a) {
    const float *w = prm + po[8] + (size_t)a * H;
    float s = 0.0f;
    for (int j = 0; j < H; ++j)
      s += w[j] * h[j];
    logits[a] = s + prm[po[9] + a];
  }
  {
    const float *w = prm + po[10];
    float s = 0.0f;
    for (int j = 0; j < H; ++j)
      s += w[j] * h[j];
    *value = s + prm[po[11]];
  }
}

#define COLS_MAX (256 * 400)

void oracle_net_forward(const float *params, int H, int A, const uint8_t *obs, int N, float *logits, float *values,
                        float *acts) {
  size_t po[13];
  oracle_param_offsets(H, A, po);
  const size_t aps = oracle_acts_per_sample(H);
#pragma omp parallel
  {
    float *cols = (float *)malloc(sizeof(float) * COLS_MAX);
    float *tmp = acts ? NULL : (float *)malloc(sizeof(float) * aps);
#pragma omp for schedule(static)
    for (int n = 0; n < N; ++n)
      forward_one(params, po, H, A, obs + (size_t)n * X0, logits + (size_t)n * A, values + n,
                  acts ? acts + (size_t)n * aps : tmp, cols);
    free(cols);
    free(tmp);
  }
}

/* dW[oc][k] += sum_p dz[oc][p] cols[k][p];  db[oc] += sum_p dz[oc][p] */
static void conv_wgrad(const float *dz, const float *cols, int OC, int Kd, int P, float *dw, float *db) {
  for (int oc = 0; oc < OC; ++oc) {
    const float *d = dz + (size_t)oc * P;
    float sb = 0.0f;
    for (int p = 0; p < P; ++p)
      sb += d[p];
    db[oc] += sb;
    for (int k = 0; k < Kd; ++k) {
      const float *c = cols + (size_t)k * P;
      float s = 0.0f;
      for (int p = 0; p < P; ++p)
        s += d[p] * c[p];
      dw[(size_t)oc * Kd + k] += s;
    }
  }
}
/* dcols[k][p] = sum_oc w[oc][k] dz[oc][p] */
static void conv_dcols(const float *w, const float *dz, int OC, int Kd, int P, float *dcols) {
  memset(dcols, 0, sizeof(float) * (size_t)Kd * P);
  for (int oc = 0; oc < OC; ++oc) {
    const float *d = dz + (size_t)oc * P;
    for (int k = 0; k < Kd; ++k) {
      const float wk = w[(size_t)oc * Kd + k];
      float *c = dcols + (size_t)k * P;
      for (int p = 0; p < P; ++p)
        c[p] += wk * d[p];
    }
  }
}

static void backward_one(const float *prm, const size_t *po, int H, int A, const float *acts, const float *dlogits,
                         float dvalue, float *g, float *cols, float *dcols, float *scratch) {
  const float *x0 = acts, *a1 = x0 + X0, *a2 = a1 + A1, *a3 = a2 + A2, *h = a3 + A3;
  float *dh = scratch, *da3 = dh + H, *da2 = da3 + A3, *da1 = da2 + A2;
  /* heads */
  for (int j = 0; j < H; ++j)
    dh[j] = dvalue * prm[po[10] + j];
  for (int a = 0; a < A; ++a) {
    const float d = dlogits[a];
    for (int j = 0; j < H; ++j) {
      dh[j] += d * prm[po[8] + (size_t)a * H + j];
      g[po[8] + (size_t)a * H + j] += d * h[j];
    }
    g[po[9] + a] += d;
  }
  for (int j = 0; j < H; ++j)
    g[po[10] + j] += dvalue * h[j];
  g[po[11]] += dvalue;
  /* fc */
  memset(da3, 0, sizeof(float) * A3);
  for (int o = 0; o < H; ++o) {
    const float d = dh[o];
    const float *w = prm + po[6] + (size_t)o * A3;
    float *gw = g + po[6] + (size_t)o * A3;
    for (int j = 0; j < A3; ++j) {
      da3[j] += d * w[j];
      gw[j] += d * a3[j];
    }
    g[po[7] + o] += d;
  }
  /* conv3 */
  for (int j = 0; j < A3; ++j)
    da3[j] = a3[j] > 0.0f ? da3[j] : 0.0f;
  im2col(a2, 64, 9, 9, 3, 1, 7, 7, cols);
  conv_wgrad(da3, cols, 64, 576, 49, g + po[4], g + po[5]);
  conv_dcols(prm + po[4], da3, 64, 576, 49, dcols);
  memset(da2, 0, sizeof(float) * A2);
  col2im_add(dcols, 64, 9, 9, 3, 1, 7, 7, da2);
  /* conv2 */
  for (int j = 0; j < A2; ++j)
    da2[j] = a2[j] > 0.0f ? da2[j] : 0.0f;
  im2col(a1, 32, 20, 20, 4, 2, 9, 9, cols);
  conv_wgrad(da2, cols, 64, 512, 81, g + po[2], g + po[3]);
  conv_dcols(prm + po[2], da2, 64, 512, 81, dcols);
  memset(da1, 0, sizeof(float) * A1);
  col2im_add(dcols, 32, 20, 20, 4, 2, 9, 9, da1);
  /* conv1 (no input gradient) */
  for (int j = 0; j < A1; ++j)
    da1[j] = a1[j] > 0.0f ? da1[j] : 0.0f;
  im2col(x0, 4, 84, 84, 8, 4, 20, 20, cols);
  conv_wgrad(da1, cols, 32, 256, 400, g + po[0], g + po[1]);
}

void oracle_net_backward(const float *params, int H, int A, int N, const float *acts, const float *dlogits,
                         const float *dvalues, float *grads) {
  size_t po[13];
  oracle_param_offsets(H, A, po);
  const size_t np = po[12], aps = oracle_acts_per_sample(H);
  const int nt = oracle_num_threads();
  float *part = (float *)calloc((size_t)nt * np, sizeof(float));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    float *cols = (float *)malloc(sizeof(float) * COLS_MAX);
    float *dcols = (float *)malloc(sizeof(float) * COLS_MAX);
    float *scratch = (float *)malloc(sizeof(float) * ((size_t)H + A3 + A2 + A1));
#pragma omp for schedule(static)
    for (int n = 0; n < N; ++n)
      backward_one(params, po, H, A, acts + (size_t)n * aps, dlogits + (size_t)n * A, dvalues[n],
                   part + (size_t)tid * np, cols, dcols, scratch);
    free(cols);
    free(dcols);
    free(scratch);
  }
  for (size_t i = 0; i < np; ++i) { /* fixed thread order -> deterministic for a fixed thread count */
    float s = 0.0f;
    for (int t = 0; t < nt; ++t)
      s += part[(size_t)t * np + i];
    grads[i] = s;
  }
  free(part);
}

/* ------------------------------------------------------------------ PPO loss: src/ai/ppo/losses.cc:4-43 */
float oracle_ppo_loss(const float *logits, const float *old_logp, const int64_t *actions, const float *adv,
                      const float *values, const float *returns, const uint8_t *masks, int B, int A, float clip,
                      float c_v, float c_e, float n_mask, float *clipped, float *value_losses, float *entropies,
                      float *total_losses, float *ratio, float *dlogits, float *dvalues) {
  float nm = n_mask;
  if (nm <= 0.0f) {
    nm = 0.0f;
    for (int i = 0; i < B; ++i)
      nm += masks[i] ? 1.0f : 0.0f; /* losses.cc:19 masks.sum() */
  }
  float lp[64], p[64];
  double loss = 0.0;
  for (int i = 0; i < B; ++i) {
    oracle_log_softmax(logits + (size_t)i * A, lp, 1, A); /* train.h:119 normalize_logits */
    const int a = (int)actions[i];
    float ent = 0.0f;
    for (int k = 0; k < A; ++k) {
      p[k] = expf(lp[k]);
      ent += p[k] * lp[k]; /* losses.cc:41-43 */
    }
    ent = -ent;
    const float rho = expf(lp[a] - old_logp[(size_t)i * A + a]); /* losses.cc:33 */
    float crho = rho < 1.0f - clip ? 1.0f - clip : (rho > 1.0f + clip ? 1.0f + clip : rho);
    const float un = rho * adv[i], cl = crho * adv[i];
    const float obj = un < cl ? un : cl;                         /* losses.cc:38 */
    const float dv = values[i] - returns[i];
    const float lv = 0.5f * (dv * dv);                           /* losses.cc:15 */
    const float L = -obj + c_v * lv - c_e * ent;                 /* losses.cc:17-18 */
    if (masks[i])
      loss += L;
    if (clipped) clipped[i] = obj;
    if (value_losses) value_losses[i] = lv;
    if (entropies) entropies[i] = ent;
    if (total_losses) total_losses[i] = L;
    if (ratio) ratio[i] = rho;
    if (dlogits) { /* closed form, SURVEY app. B (verified there against autograd) */
      const float m = masks[i] ? 1.0f / nm : 0.0f;
      const int active = adv[i] >= 0.0f ? (rho <= 1.0f + clip) : (rho >= 1.0f - clip);
      const float gs = active ? -rho * adv[i] : 0.0f;
      for (int k = 0; k < A; ++k)
        dlogits[(size_t)i * A + k] = m * (gs * ((k == a ? 1.0f : 0.0f) - p[k]) + c_e * p[k] * (lp[k] + ent));
      dvalues[i] = m * c_v * dv;
    }
  }
  return (float)(loss / nm);
}

/* ------------------------------------------------------------------ clip + Adam */
/* src/ai/ppo/train.cc:12-46 */
float oracle_clip_grad_norm(float *grads, int H, int A, float max_norm) {
  size_t po[13];
  oracle_param_offsets(H, A, po);
  double tot = 0.0;
  for (int k = 0; k < 12; ++k) { /* norm of the per-tensor norms, :32-37 */
    double s = 0.0;
    for (size_t i = po[k]; i < po[k + 1]; ++i)
      s += (double)grads[i] * grads[i];
    const float nk = (float)sqrt(s);
    tot += (double)nk * nk;
  }
  const float total = (float)sqrt(tot);
  float coef = max_norm / (total + 1e-6f); /* :39 */
  if (coef > 1.0f)
    coef = 1.0f;                           /* :40-41 */
  for (size_t i = 0; i < po[12]; ++i)
    grads[i] *= coef;                      /* :42-44 always applied */
  return total;
}

/* torch::optim::Adam (eps 1e-5, src/bin/train.cc:360-362), form checked against the compiled reference */
void oracle_adam_step(float *p, const float *g, float *m, float *v, size_t n, double lr, double beta1, double beta2,
                      double eps, int64_t step) {
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2s = (float)sqrt(bc2);
  const float b1 = (float)beta1, b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
  const float e = (float)eps;
  for (size_t i = 0; i < n; ++i) {
    m[i] = m[i] * b1 + omb1 * g[i];
    v[i] = v[i] * b2 + omb2 * (g[i] * g[i]);
    const float denom = sqrtf(v[i]) / bc2s + e;
    p[i] = p[i] - step_size * (m[i] / denom);
  }
}

/* ------------------------------------------------------------------ train loop: src/ai/ppo/train.h:114-157 */
int oracle_train(float *params, float *adam_m, float *adam_v, int64_t *adam_step, int H, int A, const uint8_t *obs,
                 const int64_t *actions, const float *old_logp, const float *adv, const float *returns,
                 const uint8_t *masks, int N, int epochs, int M, double lr, float clip, float c_v, float c_e,
                 float max_norm, float *loss, float *grad_norm, float *total_losses, float *ratio, float *entropies,
                 float *value_losses, float *clipped, float *last_grads) {
  if (N % M != 0)
    return -1; /* train.h:140-143 */
  const int B = N / M;
  const size_t np = oracle_param_count(H, A), aps = oracle_acts_per_sample(H);
  float *acts = (float *)malloc(sizeof(float) * aps * B);
  float *logits = (float *)malloc(sizeof(float) * (size_t)B * A);
  float *values = (float *)malloc(sizeof(float) * B);
  float *dlogits = (float *)malloc(sizeof(float) * (size_t)B * A);
  float *dvalues = (float *)malloc(sizeof(float) * B);
  float *grads = (float *)malloc(sizeof(float) * np);
  for (int ep = 0; ep < epochs; ++ep)
    for (int k = 0; k < M; ++k) { /* contiguous slices, randperm unused (Q1) */
      const size_t s = (size_t)k * B, mi = (size_t)ep * M + k;
      oracle_net_forward(params, H, A, obs + s * X0, B, logits, values, acts);
      loss[mi] = oracle_ppo_loss(
          logits, old_logp + s * A, actions + s, adv + s, values, returns + s, masks + s, B, A, clip, c_v, c_e, 0.0f,
          clipped ? clipped + mi * B : NULL, value_losses ? value_losses + mi * B : NULL,
          entropies ? entropies + mi * B : NULL, total_losses ? total_losses + mi * B : NULL,
          ratio ? ratio + mi * B : NULL, dlogits, dvalues);
      oracle_net_backward(params, H, A, B, acts, dlogits, dvalues, grads);
      grad_norm[mi] = oracle_clip_grad_norm(grads, H, A, max_norm);
      *adam_step += 1;
      oracle_adam_step(params, grads, adam_m, adam_v, np, lr, 0.9, 0.999, 1e-5, *adam_step);
    }
  if (last_grads)
    memcpy(last_grads, grads, sizeof(float) * np);
  free(acts);
  free(logits);
  free(values);
  free(dlogits);
  free(dvalues);
  free(grads);
  return 0;
}

/* ------------------------------------------------------------------ advantage normalisation
 * NOT in the reference (grep over src/ finds no advantage normalisation) - BASELINE.json's north_star
 * lists it, the library offers it as an off-by-default extension.  PARITY UNPINNED: no reference function, test or
 * fixture covers it; this is the textbook definition the extension follows: over the unmasked samples
 * (masks = !episode_starts, src/ai/buffer.cc:74) adv <- (adv - mean) / (std + 1e-8) with the unbiased standard
 * deviation (torch.Tensor.std default); masked samples are transformed too (they never enter the loss).
 * Accumulates in double: it is the checker.  returns the unmasked count. */
long oracle_adv_norm(float *adv, const uint8_t *masks, long n) {
  double s = 0.0, q = 0.0;
  long c = 0;
  for (long i = 0; i < n; ++i)
    if (masks[i]) {
      s += adv[i];
      q += (double)adv[i] * adv[i];
      ++c;
    }
  if (c == 0)
    return 0;
  const double mean = s / (double)c;
  double var = (q - (double)c * mean * mean) / (double)(c > 1 ? c - 1 : 1);
  if (var < 0.0)
    var = 0.0;
  const double inv = 1.0 / (sqrt(var) + 1e-8);
  for (long i = 0; i < n; ++i)
    adv[i] = (float)(((double)adv[i] - mean) * inv);
  return c;
}
