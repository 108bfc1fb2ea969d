// TEST INFRASTRUCTURE - a host-only stand-in for libaleppo.so that lets the trainer shell's HOST side (worker pool,
// slot protocol, mapped frame buffer hand-off, RANK / WORLD_SIZE rendezvous) run without a GPU, under ThreadSanitizer
// ("TSan-clean handoff"; the reference's own loop has a benign race at src/ai/rollout.cc:303-313).
// It computes nothing: actions are a counter pattern, metrics are constants.  What it does do is touch the caller's
// buffers exactly where the real library's device work would - frames and episode-start bytes are READ when the step is
// released (aleppo_release_step / aleppo_step: the ingest kernel), the action buffer is WRITTEN in aleppo_act - so that
// TSan sees every cross-thread hand-off the trainer relies on.  Never linked into anything but trainer/train_tsan.
#include "../../include/aleppo.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

struct aleppo_ctx {
  aleppo_config cfg{};
  std::vector<int64_t> actions;
  std::vector<float> params;
  const uint8_t *armed_frames = nullptr, *armed_start = nullptr;
  int armed_kind = 0;
  bool armed = false;
  int t = 0;
  uint64_t tick = 0, checksum = 0;
  std::string err;
  size_t nparams() const {
    const size_t H = (size_t)cfg.hidden_size, A = (size_t)cfg.num_actions;
    return 32 * 256 + 32 + 64 * 512 + 64 + 64 * 576 + 64 + H * 3136 + H + A * H + A + H + 1;
  }
  void ingest(const uint8_t *frames, int kind, const uint8_t *start) { // what the ingest kernel would read
    const size_t per = kind == ALEPPO_FRAMES_RAW_PAIR ? 2 * 210 * 160 : 84 * 84;
    for (size_t i = 0; i < (size_t)cfg.num_envs * per; i += 97)
      checksum = checksum * 31 + frames[i];
    for (int e = 0; e < cfg.num_envs; ++e)
      checksum += start[e];
  }
};
static thread_local std::string g_err;
static int fail(aleppo_ctx *c, int code, const char *m) {
  (c ? c->err : g_err) = m;
  return code;
}
extern "C" {
int aleppo_abi_version(void) { return ALEPPO_ABI_VERSION; }
const char *aleppo_last_error(const aleppo_ctx *c) { return c ? c->err.c_str() : g_err.c_str(); }
int aleppo_create(const aleppo_config *cfg, aleppo_ctx **out) {
  if (!cfg || !out || cfg->abi_version != ALEPPO_ABI_VERSION)
    return fail(nullptr, ALEPPO_ERR_INVALID_ARGUMENT, "stub: bad config");
  aleppo_ctx *c = new aleppo_ctx();
  c->cfg = *cfg;
  c->actions.assign((size_t)cfg->num_envs, 0);
  c->params.assign(c->nparams(), 0.f);
  *out = c;
  return ALEPPO_OK;
}
void aleppo_destroy(aleppo_ctx *c) { delete c; }
int aleppo_param_count(const aleppo_ctx *c, size_t *n) {
  *n = c->nparams();
  return ALEPPO_OK;
}
int aleppo_load_params(aleppo_ctx *c, const float *p, size_t n) {
  if (n != c->params.size())
    return fail(c, ALEPPO_ERR_INVALID_ARGUMENT, "stub: wrong count");
  std::memcpy(c->params.data(), p, n * 4);
  return ALEPPO_OK;
}
int aleppo_export_params(aleppo_ctx *c, float *p, size_t n) {
  if (n != c->params.size())
    return fail(c, ALEPPO_ERR_INVALID_ARGUMENT, "stub: wrong count");
  std::memcpy(p, c->params.data(), n * 4);
  return ALEPPO_OK;
}
int aleppo_act(aleppo_ctx *c, const float *, const int64_t **out) {
  if (c->armed)
    return fail(c, ALEPPO_ERR_RUNTIME, "a step is armed: call aleppo_release_step first");
  for (size_t e = 0; e < c->actions.size(); ++e)
    c->actions[e] = (int64_t)((c->tick * 7 + e * 3) % (uint64_t)c->cfg.num_actions);
  c->tick++;
  *out = c->actions.data();
  return ALEPPO_OK;
}
int aleppo_host_alloc(aleppo_ctx *, size_t bytes, void **p) {
  *p = std::calloc(1, bytes);
  return *p ? ALEPPO_OK : ALEPPO_ERR_RUNTIME;
}
int aleppo_host_free(aleppo_ctx *, void *p) {
  std::free(p);
  return ALEPPO_OK;
}
int aleppo_set_gray_lut(aleppo_ctx *, const uint8_t *) { return ALEPPO_OK; }
int aleppo_step(aleppo_ctx *c, const uint8_t *frames, int kind, int, const float *r, const uint8_t *te, const uint8_t *tr,
                const uint8_t *st) {
  if (c->armed || c->t >= c->cfg.horizon)
    return fail(c, ALEPPO_ERR_RUNTIME, "stub: step out of order");
  c->ingest(frames, kind, st);
  for (int e = 0; e < c->cfg.num_envs; ++e)
    c->checksum += (uint64_t)r[e] + te[e] + tr[e];
  c->t++;
  return ALEPPO_OK;
}
int aleppo_arm_step(aleppo_ctx *c, const uint8_t *frames, int kind, const uint8_t *start, const float *) {
  if (c->armed || c->t >= c->cfg.horizon)
    return fail(c, ALEPPO_ERR_RUNTIME, "stub: arm out of order");
  c->armed_frames = frames; // NOT read here: the emulators have not produced them yet
  c->armed_start = start;
  c->armed_kind = kind;
  c->armed = true;
  return ALEPPO_OK;
}
int aleppo_release_step(aleppo_ctx *c, const float *r, const uint8_t *te, const uint8_t *tr) {
  if (!c->armed)
    return fail(c, ALEPPO_ERR_RUNTIME, "aleppo_release_step without an armed step");
  c->ingest(c->armed_frames, c->armed_kind, c->armed_start); // the released stream reads them now
  for (int e = 0; e < c->cfg.num_envs; ++e)
    c->checksum += (uint64_t)r[e] + te[e] + tr[e];
  c->armed = false;
  c->t++;
  return ALEPPO_OK;
}
int aleppo_finish_rollout(aleppo_ctx *c, const float *) {
  if (c->armed || c->t != c->cfg.horizon)
    return fail(c, ALEPPO_ERR_RUNTIME, "Buffer is not full, cannot compute GAE.");
  c->t = 0;
  return ALEPPO_OK;
}
int aleppo_train(aleppo_ctx *c, double lr, int epochs, int M, aleppo_minibatch_metrics *out) {
  for (float &p : c->params)
    p -= (float)lr * 0.5f;
  for (int i = 0; out && i < epochs * M; ++i)
    out[i] = aleppo_minibatch_metrics{0.1f, 1.0f, -0.01f, 0.2f, 1.3f, 1.0f, 1.0f};
  return ALEPPO_OK;
}
int aleppo_read_train_metric(aleppo_ctx *, int, float *dst, size_t n) {
  std::memset(dst, 0, n * 4);
  return ALEPPO_OK;
}
int aleppo_read_batch(aleppo_ctx *, int field, void *dst, size_t bytes) {
  std::memset(dst, field == ALEPPO_F_MASKS ? 1 : 0, bytes);
  return ALEPPO_OK;
}
int aleppo_set_option(aleppo_ctx *, int, int) { return ALEPPO_OK; }
int aleppo_profile_enable(aleppo_ctx *, int) { return ALEPPO_OK; }
int aleppo_profile_read(aleppo_ctx *, int, double *ms, int64_t *n) {
  *ms = 0;
  *n = 0;
  return ALEPPO_OK;
}
// the id is unique per call (pid + clock), like ncclGetUniqueId; aleppo_comm_init prints what it received so that a test
// can see that every rank of ONE launch got rank 0's id of THAT launch
int aleppo_comm_unique_id(uint8_t id[ALEPPO_UNIQUE_ID_BYTES]) {
  const uint64_t a = (uint64_t)getpid(),
                 b = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  for (int i = 0; i < ALEPPO_UNIQUE_ID_BYTES; ++i)
    id[i] = (uint8_t)((a * 0x9E3779B97F4A7C15ull + b * (uint64_t)(i + 1)) >> ((i % 7) * 8));
  return ALEPPO_OK;
}
int aleppo_comm_init(aleppo_ctx *c, const uint8_t id[ALEPPO_UNIQUE_ID_BYTES]) {
  char hex[33];
  for (int i = 0; i < 16; ++i)
    std::snprintf(hex + 2 * i, 3, "%02x", id[i]);
  std::printf("stub comm_init rank %d of %d id %s\n", c->cfg.rank, c->cfg.world_size, hex);
  std::fflush(stdout);
  return ALEPPO_OK;
}
}
