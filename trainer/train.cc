// train - C++ trainer shell over libaleppo.so (C ABI only; no libtorch, no HIP headers).
//
// Keeps the command line and the configs/*.yaml keys of the reference trainer
//   train <rom> <log path> <video dir> <group> <config.yaml> [profile]          (src/bin/train.cc:323-335)
// and re-hosts what main() and Rollout's HOST half do around the hot path:
//   * Config / load_config with the reference's keys and defaults           (src/bin/train.cc:33-63,108-136)
//   * worker threads stepping environments, fed by an index queue           (src/ai/rollout.cc:280-328, queue.h)
//     - actions are read from the pinned buffer aleppo_act returns instead of tensor.item()  (rollout.cc:312-313)
//   * the slot protocol: episode-start slots, stale rewards, flag bookkeeping, episode / game statistics,
//     total_steps counting only non-start slots                              (src/ai/rollout.cc:204-267)
//   * warm rollout before the loop, linear lr anneal, "Rollout i of N", scalar logging
//                                                                            (src/bin/train.cc:391-458,163-210)
//   * orthogonal init with gains sqrt(2) / 0.01 / 1, zero biases            (src/bin/train.cc:212-253)
//   * a TensorBoard event file (TFRecord + hand-encoded protobuf): every scalar and histogram of log_data
//     (src/bin/train.cc:163-210) and the hparams session record of logger.add_hparams (:72-105, :389)
//   * the optional 6th argument [profile] (src/bin/train.cc:409-419, 459-462 save a Kineto trace there): a
//     chrome://tracing / Perfetto JSON of every C-ABI call (host spans) plus the per-kernel-class device times the
//     library measures with HIP events; the same spans are roctx ranges (rocprofv3 --marker-trace) when libroctx64.so
//     is loadable
//   * the emulator threads write their frames into ONE page-locked, GPU-mapped buffer (aleppo_host_alloc) that the
//     ingest kernel reads in place (ALEPPO_HOST_MAPPED) - no per-slot staging copy (rollout.cc:325-326 memcpy's into
//     per-env host vectors that update_observations then stacks and uploads)
// ALE is not available in this build environment (no headers, no ROMs): the emulator behind the
// VirtualEnvironment-like interface is a deterministic synthetic Atari-shaped game (84x84 gray frames,
// 5 lives, reward on "brick hits", terminal on life loss like EpisodeLife, truncation at max_steps /
// max_return).  The rom argument is accepted and recorded but not opened.
// New OPTIONAL yaml keys (defaults reproduce the reference): precision: fp32|bf16, rollout_precision: fp32|fp16,
// device_preprocess: false (true: the emulators hand over RAW 210x160 frame pairs and the device does gray LUT + resize +
// max), advantage_norm: false, action_size (honoured here; the reference hard-codes 4, Q4), seed,
// slot_ahead: true (the next slot's ingest + acting kernels are enqueued BEFORE the emulator threads run, behind a stream
// wait that aleppo_release_step lifts when they are done: aleppo_arm_step in include/aleppo.h; false: aleppo_step).
// Data parallelism (no reference counterpart): start one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK
// in the environment (torchrun / mpirun style).  Rank r owns the contiguous environment block
// [r * E / W, (r + 1) * E / W) and GPU LOCAL_RANK; rank 0 creates the RCCL id, hands it to the others through the file
// <log path>.rcclid, and is the only rank that writes the event file (its own environments' episode statistics, the
// global - all-reduced - update metrics).  Everything else is unchanged: aleppo_train all-reduces the gradients.
#include "../include/aleppo.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <mutex>
#include <numeric>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------ config
struct Config {
  size_t total_environments = 512, hidden_size = 512, action_size = 4, horizon = 128, max_steps = 108000,
         frame_stack = 4;
  double learning_rate = 2.5e-4;
  float clip_param = 0.1f, value_loss_coef = 0.5f, entropy_coef = 0.01f;
  long num_epochs = 1, mini_batch_size = 2048, num_mini_batches = 32;
  float gae_discount = 0.99f, gae_lambda = 0.95f, max_gradient_norm = 0.5f;
  size_t num_rollouts = 7000, num_workers = 16, worker_batch_size = 32, frame_skip = 4;
  float max_return = -1.0f;
  bool record_observation = false, record_video = false, cuda_graph = false, deterministic = false;
  // extensions
  std::string precision = "fp32", rollout_precision = "fp32";
  bool device_preprocess = false; // emulators hand over raw frame pairs; gray LUT + resize + max run on the device (N2)
  bool slot_ahead = true;         // aleppo_arm_step / aleppo_release_step: the stream runs one slot ahead of the emulators
  bool advantage_norm = false;
  uint64_t seed = 42;
};

static std::string trim(const std::string &s) {
  const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}
// flat "key: value" YAML (what configs/*.yaml use): comments, blank lines, scalars
static std::map<std::string, std::string> parse_yaml(const std::string &path) {
  std::ifstream f(path);
  if (!f)
    throw std::runtime_error("cannot open config: " + path);
  std::map<std::string, std::string> kv;
  std::string line;
  while (std::getline(f, line)) {
    const size_t h = line.find('#');
    if (h != std::string::npos)
      line = line.substr(0, h);
    const size_t c = line.find(':');
    if (c == std::string::npos)
      continue;
    const std::string k = trim(line.substr(0, c)), v = trim(line.substr(c + 1));
    if (!k.empty() && !v.empty())
      kv[k] = v;
  }
  return kv;
}
template <class T> static T as(const std::map<std::string, std::string> &kv, const char *k, T dflt) {
  auto it = kv.find(k);
  if (it == kv.end())
    return dflt;
  std::istringstream ss(it->second);
  T v;
  ss >> v;
  if (ss.fail())
    throw std::runtime_error(std::string("bad value for ") + k);
  return v;
}
static bool as_bool(const std::map<std::string, std::string> &kv, const char *k, bool dflt) {
  auto it = kv.find(k);
  if (it == kv.end())
    return dflt;
  return it->second == "true" || it->second == "True" || it->second == "1" || it->second == "yes";
}
static Config load_config(const std::string &path) { // keys / defaults of src/bin/train.cc:108-136
  const auto kv = parse_yaml(path);
  Config c;
  c.total_environments = as<size_t>(kv, "total_environments", 512);
  c.hidden_size = as<size_t>(kv, "hidden_size", 512);
  c.action_size = as<size_t>(kv, "action_size", 4);
  c.horizon = as<size_t>(kv, "horizon", 128);
  c.max_steps = as<size_t>(kv, "max_steps", 108000);
  c.frame_stack = as<size_t>(kv, "frame_stack", 4);
  c.learning_rate = as<double>(kv, "learning_rate", 2.5e-4);
  c.clip_param = as<float>(kv, "clip_param", 0.1f);
  c.value_loss_coef = as<float>(kv, "value_loss_coef", 0.5f);
  c.entropy_coef = as<float>(kv, "entropy_coef", 0.01f);
  c.num_epochs = as<long>(kv, "num_epochs", 1);
  c.mini_batch_size = as<long>(kv, "mini_batch_size", 2048);
  c.num_mini_batches = as<long>(kv, "num_mini_batches", 32);
  c.gae_discount = as<float>(kv, "gae_discount", 0.99f);
  c.gae_lambda = as<float>(kv, "gae_lambda", 0.95f);
  c.max_gradient_norm = as<float>(kv, "max_gradient_norm", 0.5f);
  c.num_rollouts = as<size_t>(kv, "num_rollouts", 7000);
  c.num_workers = as<size_t>(kv, "num_workers", 16);
  c.worker_batch_size = as<size_t>(kv, "worker_batch_size", 32);
  c.frame_skip = as<size_t>(kv, "frame_skip", 4);
  c.max_return = as<float>(kv, "max_return", -1.0f);
  c.record_observation = as_bool(kv, "record_observation", false);
  c.record_video = as_bool(kv, "record_video", false);
  c.cuda_graph = as_bool(kv, "cuda_graph", false);
  c.deterministic = as_bool(kv, "deterministic", false);
  c.precision = as<std::string>(kv, "precision", "fp32");
  c.rollout_precision = as<std::string>(kv, "rollout_precision", "fp32");
  c.device_preprocess = as_bool(kv, "device_preprocess", false);
  c.slot_ahead = as_bool(kv, "slot_ahead", true);
  c.advantage_norm = as_bool(kv, "advantage_norm", false);
  c.seed = as<uint64_t>(kv, "seed", 42);
  return c;
}

// ------------------------------------------------------------------ synthetic emulator (stands in for the ALE wrapper chain)
struct StepOut {
  float reward = 0.f;
  bool terminated = false, truncated = false, game_over = false;
};
class SyntheticAtari {
public:
  // raw = true: the emulator hands over what ALE itself produces - the last TWO 210x160 palette-code frames of the skip
  // window - and the gray LUT, the 84x84 resize and the 2-frame max (environment.cc:48-55, resize.cc:34-41,
  // max_and_skip.cc:33-42) run on the device (ALEPPO_FRAMES_RAW_PAIR); raw = false: one finished 84x84 gray frame.
  SyntheticAtari(uint64_t seed, size_t max_steps, float max_return, size_t actions, bool raw = false)
      : rng_(seed * 0x9E3779B97F4A7C15ull + 12345), max_steps_(max_steps), max_return_(max_return), actions_(actions),
        raw_(raw) {}
  static size_t frame_bytes(bool raw) { return raw ? 2 * 210 * 160 : 84 * 84; }
  // FireReset / EpisodeLife semantics: a full reset only after game over, otherwise continue with the next life
  void reset(uint8_t *frame) {
    if (lives_ == 0) {
      lives_ = 5;
      steps_ = 0;
      episode_return_ = 0.f;
      bricks_ = 0;
    }
    ball_x_ = 42;
    ball_y_ = 60;
    prev_x_ = ball_x_;
    prev_y_ = ball_y_;
    dx_ = (next() & 1) ? 1 : -1;
    dy_ = -1;
    render(frame);
  }
  StepOut step(int action, uint8_t *frame) {
    StepOut o;
    paddle_ += (action == 2 ? 3 : action == 3 ? -3 : 0); // NOOP FIRE RIGHT LEFT like Breakout's minimal set
    paddle_ = std::clamp(paddle_, 4, 79);
    for (int k = 0; k < 4; ++k) { // frame_skip emulator frames per agent step
      prev_x_ = ball_x_;
      prev_y_ = ball_y_;
      ball_x_ += dx_ * 2;
      ball_y_ += dy_ * 2;
      if (ball_x_ <= 1 || ball_x_ >= 82)
        dx_ = -dx_;
      if (ball_y_ <= 20) { // brick row
        dy_ = 1;
        o.reward += (float)(1 + 3 * (bricks_ % 3 == 2));
        ++bricks_;
      }
      if (ball_y_ >= 78) {
        if (std::abs(ball_x_ - paddle_) <= 8 || (next() % 3) == 0)
          dy_ = -1;
        else { // life lost -> EpisodeLife reports a terminal
          --lives_;
          o.terminated = true;
          break;
        }
      }
    }
    steps_ += 4;
    episode_return_ += o.reward;
    o.game_over = lives_ == 0;
    if (!o.terminated && (steps_ >= max_steps_ || (max_return_ > 0 && episode_return_ >= max_return_))) {
      o.truncated = true; // ALE max_num_frames_per_episode / TruncateOnEpisodeReturn
      lives_ = 0;
      o.game_over = true;
    }
    render(frame);
    (void)actions_;
    return o;
  }

private:
  uint64_t next() {
    rng_ ^= rng_ << 13;
    rng_ ^= rng_ >> 7;
    rng_ ^= rng_ << 17;
    return rng_;
  }
  void render(uint8_t *f) const {
    if (raw_) { // two emulator frames (the ball at its previous and current position), ALE-style even palette codes
      for (int k = 0; k < 2; ++k) {
        uint8_t *g = f + (size_t)k * 210 * 160;
        std::memset(g, 0, 210 * 160);
        auto rect = [&](int x0, int x1, int y0, int y1, uint8_t c) { // [x0,x1) x [y0,y1) in 84-grid units
          for (int y = y0 * 210 / 84; y < y1 * 210 / 84; ++y)
            for (int x = x0 * 160 / 84; x < x1 * 160 / 84; ++x)
              if (x >= 0 && x < 160 && y >= 0 && y < 210)
                g[y * 160 + x] = c;
        };
        for (int y = 8; y < 20; y += 3)
          for (int x = 0; x < 84; x += 6)
            rect(x, x + 6, y, y + 3, (uint8_t)((((x / 6 + y / 3 + bricks_) % 4) * 50 + 60) & ~1));
        rect(paddle_ - 6, paddle_ + 7, 80, 82, 200);
        const int bx = k == 0 ? prev_x_ : ball_x_, by = k == 0 ? prev_y_ : ball_y_;
        rect(bx, bx + 2, by, by + 2, 236);
      }
      return;
    }
    std::memset(f, 0, 84 * 84);
    for (int y = 8; y < 20; ++y)
      for (int x = 0; x < 84; ++x)
        f[y * 84 + x] = (uint8_t)(((x / 6 + y / 3 + bricks_) % 4) * 50 + 60);
    for (int x = paddle_ - 6; x <= paddle_ + 6; ++x)
      if (x >= 0 && x < 84)
        f[80 * 84 + x] = f[81 * 84 + x] = 200;
    for (int y = ball_y_; y < ball_y_ + 2; ++y)
      for (int x = ball_x_; x < ball_x_ + 2; ++x)
        if (x >= 0 && x < 84 && y >= 0 && y < 84)
          f[y * 84 + x] = 236;
  }
  uint64_t rng_;
  size_t max_steps_;
  float max_return_;
  size_t actions_;
  bool raw_;
  int lives_ = 0, paddle_ = 42, ball_x_ = 42, ball_y_ = 60, prev_x_ = 42, prev_y_ = 60, dx_ = 1, dy_ = -1, bricks_ = 0;
  size_t steps_ = 0;
  float episode_return_ = 0.f;
};

// ------------------------------------------------------------------ worker pool (std::thread + index queue, rollout.cc:280-297)
class WorkerPool {
public:
  WorkerPool(size_t n, std::function<void(size_t)> fn) : fn_(std::move(fn)) {
    for (size_t i = 0; i < n; ++i)
      threads_.emplace_back([this] { loop(); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : threads_)
      t.join();
  }
  void run_all(size_t count) { // push indices 0..count-1, wait until all are done (step_all)
    {
      std::lock_guard<std::mutex> l(m_);
      next_ = 0;
      end_ = count;
      done_ = 0;
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> l(m_);
    done_cv_.wait(l, [&] { return done_ == end_; });
  }

private:
  void loop() {
    for (;;) {
      size_t i;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || next_ < end_; });
        if (stop_)
          return;
        i = next_++;
      }
      fn_(i);
      {
        std::lock_guard<std::mutex> l(m_);
        if (++done_ == end_)
          done_cv_.notify_all();
      }
    }
  }
  std::function<void(size_t)> fn_;
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  size_t next_ = 0, end_ = 0, done_ = 0;
  bool stop_ = false;
};

// ------------------------------------------------------------------ TensorBoard event file (TFRecord + protobuf by hand)
static uint32_t crc32c(const uint8_t *p, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k)
        c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i)
    c = table[(c ^ p[i]) & 255] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
static uint32_t masked_crc(const uint8_t *p, size_t n) {
  const uint32_t c = crc32c(p, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}
struct Pb { // minimal protobuf encoder
  std::string b;
  void varint(uint64_t v) {
    while (v >= 128) {
      b.push_back((char)(v | 128));
      v >>= 7;
    }
    b.push_back((char)v);
  }
  void key(int field, int wire) { varint(((uint64_t)field << 3) | wire); }
  void f64(int field, double v) {
    key(field, 1);
    b.append(reinterpret_cast<const char *>(&v), 8);
  }
  void f32(int field, float v) {
    key(field, 5);
    b.append(reinterpret_cast<const char *>(&v), 4);
  }
  void i64(int field, int64_t v) {
    key(field, 0);
    varint((uint64_t)v);
  }
  void bytes(int field, const std::string &s) {
    key(field, 2);
    varint(s.size());
    b += s;
  }
  void packed_f64(int field, const std::vector<double> &v) {
    key(field, 2);
    varint(v.size() * 8);
    b.append(reinterpret_cast<const char *>(v.data()), v.size() * 8);
  }
};
class EventWriter {
public:
  explicit EventWriter(const std::string &path) : f_(path, std::ios::binary) {
    if (!f_)
      throw std::runtime_error("cannot open event file: " + path);
    Pb e;
    e.f64(1, now());
    e.bytes(3, "brain.Event:2"); // file_version
    record(e.b);
  }
  void add_scalar(const std::string &tag, int64_t step, float value) {
    Pb v;
    v.bytes(1, tag);
    v.f32(2, value);
    Pb s;
    s.bytes(1, v.b);
    Pb e;
    e.f64(1, now());
    e.i64(2, step);
    e.bytes(5, s.b);
    record(e.b);
  }
  void add_histogram(const std::string &tag, int64_t step, const std::vector<float> &x) {
    if (x.empty())
      return;
    double mn = x[0], mx = x[0], sum = 0, sq = 0;
    for (float v : x) {
      mn = std::min<double>(mn, v);
      mx = std::max<double>(mx, v);
      sum += v;
      sq += (double)v * v;
    }
    const int nb = 30;
    std::vector<double> limits(nb), counts(nb, 0.0);
    const double w = (mx - mn) / nb > 0 ? (mx - mn) / nb : 1.0;
    for (int i = 0; i < nb; ++i)
      limits[i] = mn + w * (i + 1);
    for (float v : x)
      counts[std::min(nb - 1, (int)((v - mn) / w))] += 1.0;
    Pb h; // HistogramProto: min=1 max=2 num=3 sum=4 sum_squares=5 bucket_limit=6 bucket=7
    h.f64(1, mn);
    h.f64(2, mx);
    h.f64(3, (double)x.size());
    h.f64(4, sum);
    h.f64(5, sq);
    h.packed_f64(6, limits);
    h.packed_f64(7, counts);
    Pb v;
    v.bytes(1, tag);
    v.bytes(5, h.b); // Summary.Value.histo
    Pb s;
    s.bytes(1, v.b);
    Pb e;
    e.f64(1, now());
    e.i64(2, step);
    e.bytes(5, s.b);
    record(e.b);
  }
  // logger.add_hparams(get_parameters(config), group_name, start_time) (src/bin/train.cc:72-105, :389): the HParams
  // plugin's session-start record.  Summary.Value{tag "_hparams_/session_start_info", metadata.plugin_data{plugin_name
  // "hparams", content = HParamsPluginData{version 0, session_start_info{hparams map<string, google.protobuf.Value>,
  // group_name, start_time_secs}}}}
  void add_hparams(const std::vector<std::pair<std::string, double>> &numbers,
                   const std::vector<std::pair<std::string, bool>> &flags, const std::string &group, double start_secs) {
    Pb ssi;
    auto entry = [&](const std::string &k, const Pb &val) {
      Pb kv; // map entry: key = 1, value = 2
      kv.bytes(1, k);
      kv.bytes(2, val.b);
      ssi.bytes(1, kv.b);
    };
    for (auto &n : numbers) {
      Pb v;
      v.f64(2, n.second); // google.protobuf.Value.number_value
      entry(n.first, v);
    }
    for (auto &b : flags) {
      Pb v;
      v.i64(4, b.second ? 1 : 0); // google.protobuf.Value.bool_value
      entry(b.first, v);
    }
    ssi.bytes(4, group);
    ssi.f64(5, start_secs);
    Pb plugin; // HParamsPluginData: version = 1, session_start_info = 3
    plugin.i64(1, 0);
    plugin.bytes(3, ssi.b);
    Pb pd; // SummaryMetadata.PluginData: plugin_name = 1, content = 2
    pd.bytes(1, "hparams");
    pd.bytes(2, plugin.b);
    Pb md; // SummaryMetadata: plugin_data = 1
    md.bytes(1, pd.b);
    Pb v; // Summary.Value: tag = 1, metadata = 9
    v.bytes(1, "_hparams_/session_start_info");
    v.bytes(9, md.b);
    Pb s;
    s.bytes(1, v.b);
    Pb e;
    e.f64(1, now());
    e.bytes(5, s.b);
    record(e.b);
  }
  void flush() { f_.flush(); }

private:
  static double now() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }
  void record(const std::string &data) {
    const uint64_t len = data.size();
    const uint32_t c1 = masked_crc(reinterpret_cast<const uint8_t *>(&len), 8);
    const uint32_t c2 = masked_crc(reinterpret_cast<const uint8_t *>(data.data()), data.size());
    f_.write(reinterpret_cast<const char *>(&len), 8);
    f_.write(reinterpret_cast<const char *>(&c1), 4);
    f_.write(data.data(), (std::streamsize)data.size());
    f_.write(reinterpret_cast<const char *>(&c2), 4);
  }
  std::ofstream f_;
};

// ------------------------------------------------------------------ [profile] argument: host spans + roctx ranges
class Profile {
public:
  explicit Profile(const std::string &path) : path_(path), t0_(std::chrono::steady_clock::now()) {
    if (path_.empty())
      return;
    if (void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL)) { // optional: markers for rocprofv3 --marker-trace
      push_ = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
      pop_ = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    }
  }
  bool on() const { return !path_.empty(); }
  struct Span {
    Profile *p;
    const char *name;
    double t0;
    Span(Profile *p_, const char *n) : p(p_), name(n), t0(0) {
      if (!p->on())
        return;
      t0 = p->now_us();
      if (p->push_)
        p->push_(name);
    }
    ~Span() {
      if (!p->on())
        return;
      if (p->pop_)
        p->pop_();
      p->events_.push_back({name, t0, p->now_us() - t0});
    }
  };
  void device_summary(aleppo_ctx *ctx) { // per-kernel-class device time (HIP events on the kernels' streams)
    static const char *names[ALEPPO_K_COUNT] = {"ingest", "gae", "head", "adam", "conv1_fwd", "conv2_fwd", "conv3_fwd",
                                                "fc_fwd", "fc_dgrad", "fc_wgrad", "conv3_dgrad", "conv3_wgrad",
                                                "conv2_dgrad", "conv2_wgrad", "conv1_wgrad", "reduce", "infer_head",
                                                "act_fused", "conv_fwd", "conv_bwd"};
    for (int k = 0; k < ALEPPO_K_COUNT; ++k) {
      double ms = 0;
      int64_t n = 0;
      if (aleppo_profile_read(ctx, k, &ms, &n) == ALEPPO_OK && n > 0)
        device_.push_back({names[k], ms, n});
    }
  }
  void save() {
    if (!on())
      return;
    std::ofstream f(path_);
    f << "{\"traceEvents\": [\n";
    bool first = true;
    for (auto &e : events_) {
      f << (first ? "" : ",\n") << "{\"name\": \"" << e.name << "\", \"ph\": \"X\", \"pid\": 1, \"tid\": 1, \"ts\": "
        << e.ts << ", \"dur\": " << e.dur << "}";
      first = false;
    }
    f << "\n],\n\"device_kernel_classes\": [\n";
    first = true;
    for (auto &d : device_) {
      f << (first ? "" : ",\n") << "{\"kernel_class\": \"" << d.name << "\", \"avg_ms\": " << d.ms
        << ", \"launches\": " << d.n << "}";
      first = false;
    }
    f << "\n]}\n";
  }

private:
  friend struct Span;
  double now_us() const { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0_).count(); }
  struct Ev {
    const char *name;
    double ts, dur;
  };
  struct Dev {
    const char *name;
    double ms;
    int64_t n;
  };
  std::string path_;
  std::chrono::steady_clock::time_point t0_;
  int (*push_)(const char *) = nullptr;
  int (*pop_)() = nullptr;
  std::vector<Ev> events_;
  std::vector<Dev> device_;
};

// ------------------------------------------------------------------ orthogonal init (train.cc:212-228)
// rows x cols matrix of N(0,1), orthonormalised (modified Gram-Schmidt on the smaller dimension), times gain
static void orthogonal(float *w, size_t rows, size_t cols, double gain, std::mt19937_64 &g) {
  std::normal_distribution<double> nd(0.0, 1.0);
  const bool tr = rows < cols;
  const size_t R = tr ? cols : rows, C = tr ? rows : cols; // R >= C: orthonormal columns
  std::vector<double> a(R * C);
  for (auto &v : a)
    v = nd(g);
  for (size_t j = 0; j < C; ++j) {
    for (size_t k = 0; k < j; ++k) {
      double dot = 0;
      for (size_t i = 0; i < R; ++i)
        dot += a[i * C + j] * a[i * C + k];
      for (size_t i = 0; i < R; ++i)
        a[i * C + j] -= dot * a[i * C + k];
    }
    double n = 0;
    for (size_t i = 0; i < R; ++i)
      n += a[i * C + j] * a[i * C + j];
    n = std::sqrt(n);
    for (size_t i = 0; i < R; ++i)
      a[i * C + j] /= n;
  }
  for (size_t r = 0; r < rows; ++r)
    for (size_t c = 0; c < cols; ++c)
      w[r * cols + c] = (float)(gain * (tr ? a[c * C + r] : a[r * C + c]));
}
static std::vector<float> init_params(size_t H, size_t A, uint64_t seed) { // libtorch parameters() order
  std::mt19937_64 g(seed);
  const double s2 = std::sqrt(2.0);
  struct T {
    size_t rows, cols;
    double gain;
  };
  const T t[6] = {{32, 4 * 8 * 8, s2}, {64, 32 * 4 * 4, s2}, {64, 64 * 3 * 3, s2}, {H, 3136, s2}, {A, H, 0.01}, {1, H, 1.0}};
  std::vector<float> p;
  for (const T &x : t) {
    const size_t o = p.size();
    p.resize(o + x.rows * x.cols + x.rows, 0.0f); // weight then zero bias
    orthogonal(p.data() + o, x.rows, x.cols, x.gain, g);
  }
  return p;
}

static void check(aleppo_ctx *ctx, int rc) { // the reference throws at the same places
  if (rc == ALEPPO_OK)
    return;
  const char *m = aleppo_last_error(ctx);
  if (rc == ALEPPO_ERR_INVALID_ARGUMENT)
    throw std::invalid_argument(m ? m : "invalid argument");
  throw std::runtime_error(m ? m : "aleppo error");
}
template <class T> static float meanf(const std::vector<T> &v) {
  return v.empty() ? 0.f : (float)(std::accumulate(v.begin(), v.end(), 0.0) / (double)v.size());
}

int main(int argc, char **argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s <rom> <tensorboard log path> <video dir> <group> <config.yaml> [profile]\n", argv[0]);
    return 2;
  }
  try {
    const auto start_time = std::chrono::system_clock::now().time_since_epoch().count();
    const std::string rom_path = argv[1], group = argv[4];
    const std::string profile_path = argc > 6 ? argv[6] : ""; // train.cc:328-335
    Profile prof(profile_path);
    std::string log_path = argv[2];
    { // replace_extension("tfevents.<start-time>") like train.cc:324-325
      const size_t slash = log_path.find_last_of('/'), dot = log_path.find_last_of('.');
      if (dot != std::string::npos && (slash == std::string::npos || dot > slash))
        log_path = log_path.substr(0, dot);
      log_path += ".tfevents." + std::to_string(start_time);
    }
    const Config cfg = load_config(argv[5]);
    { // train.cc:347-352: create the log (and video) directories
      const auto parent = std::filesystem::path(log_path).parent_path();
      if (!parent.empty() && !std::filesystem::exists(parent))
        std::filesystem::create_directories(parent);
    }
    auto env_int = [](const char *k, int dflt) {
      const char *v = std::getenv(k);
      return v ? std::atoi(v) : dflt;
    };
    const int world = std::max(1, env_int("WORLD_SIZE", 1)), rank = env_int("RANK", 0), local_rank = env_int("LOCAL_RANK", rank);
    if (rank < 0 || rank >= world)
      throw std::runtime_error("RANK must be in [0, WORLD_SIZE)");
    if (world > 1) // (before the first HIP call: the runtime reads it when it initialises; see bench.py / DESIGN.md 6)
      setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
    if (cfg.total_environments % (size_t)world)
      throw std::runtime_error("total_environments must be divisible by WORLD_SIZE");
    const size_t E = cfg.total_environments / (size_t)world, T = cfg.horizon, A = cfg.action_size; // E: THIS rank's envs
    const size_t env0 = (size_t)rank * E;                                                           // its first environment
    if ((E * T) % (size_t)cfg.num_mini_batches)
      throw std::runtime_error("Batch size must be divisible by num_mini_batches");
    if (E % cfg.worker_batch_size)
      std::cerr << "warning: total_environments % worker_batch_size != 0 would deadlock the reference's queue\n";
    if (cfg.record_video)
      std::cerr << "note: record_video ignored (no ffmpeg / ALE in this build)\n";

    if (const char *dump = std::getenv("ALEPPO_TRAINER_DUMP_INIT")) { // test hook: the initial parameters, no GPU needed
      const std::vector<float> p = init_params(cfg.hidden_size, A, cfg.deterministic ? 42 : (uint64_t)start_time);
      std::ofstream f(dump, std::ios::binary);
      f.write(reinterpret_cast<const char *>(p.data()), (std::streamsize)(p.size() * sizeof(float)));
      std::cout << "initial parameters written: " << p.size() << std::endl;
      return 0;
    }
    aleppo_config ac{};
    ac.abi_version = ALEPPO_ABI_VERSION;
    ac.device_ordinal = local_rank;
    ac.world_size = world;
    ac.rank = rank;
    ac.num_envs = (int32_t)E;
    ac.horizon = (int32_t)T;
    ac.num_actions = (int32_t)A;
    ac.hidden_size = (int32_t)cfg.hidden_size;
    ac.frame_stack = (int32_t)cfg.frame_stack;
    ac.precision = cfg.precision == "bf16" ? ALEPPO_BF16 : ALEPPO_FP32;
    ac.rollout_precision = cfg.rollout_precision == "fp16" ? ALEPPO_ROLLOUT_FP16 : ALEPPO_ROLLOUT_FP32;
    ac.advantage_norm = cfg.advantage_norm;
    ac.gamma = cfg.gae_discount;
    ac.lambda = cfg.gae_lambda;
    ac.clip_param = cfg.clip_param;
    ac.value_loss_coef = cfg.value_loss_coef;
    ac.entropy_coef = cfg.entropy_coef;
    ac.max_gradient_norm = cfg.max_gradient_norm;
    ac.seed = cfg.seed;
    aleppo_ctx *ctx = nullptr;
    check(nullptr, aleppo_create(&ac, &ctx));
    std::cout << "MI355X is available! Training on GPU (rom argument '" << rom_path << "' -> synthetic emulator)."
              << std::endl;
    if (world > 1) {
      // RCCL communicator: rank 0's 128-byte id travels through a file next to the log.  The file belongs to ONE launch:
      // its name carries the launcher's MASTER_PORT (+ torchrun's run id when there is one), rank 0 removes whatever a
      // crashed earlier launch left under that name BEFORE it creates the id and removes its own file once every rank
      // has joined (aleppo_comm_init is collective), and the other ranks ignore a file older than their own start: a
      // stale id would leave ncclCommInitRank waiting for ever on mismatched ids.
      std::string nonce = std::getenv("MASTER_PORT") ? std::getenv("MASTER_PORT") : "0";
      if (const char *rid = std::getenv("TORCHELASTIC_RUN_ID"))
        nonce += std::string(".") + rid;
      const std::string idfile = std::string(argv[2]) + ".rcclid." + nonce;
      const auto proc_start = std::filesystem::file_time_type::clock::now();
      uint8_t id[ALEPPO_UNIQUE_ID_BYTES];
      if (rank == 0) {
        std::error_code ec;
        std::filesystem::remove(idfile, ec);
        for (int r = 1; r < world; ++r)
          std::filesystem::remove(idfile + ".ack." + std::to_string(r), ec);
        check(nullptr, aleppo_comm_unique_id(id));
        {
          std::ofstream f(idfile + ".tmp", std::ios::binary);
          f.write(reinterpret_cast<const char *>(id), sizeof(id));
          if (!f)
            throw std::runtime_error("cannot write " + idfile + ".tmp");
        }
        std::filesystem::rename(idfile + ".tmp", idfile); // atomic: a reader sees all 128 bytes or no file
      } else {
        const double limit = std::getenv("ALEPPO_RENDEZVOUS_TIMEOUT_S") ? std::atof(std::getenv("ALEPPO_RENDEZVOUS_TIMEOUT_S")) : 300.0;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
          std::error_code ec;
          const auto mt = std::filesystem::last_write_time(idfile, ec);
          // (ranks of one launch start within seconds of each other: anything written more than two minutes before this
          // process started is a leftover)
          if (!ec && mt + std::chrono::seconds(120) >= proc_start) {
            std::ifstream f(idfile, std::ios::binary);
            if (f && f.read(reinterpret_cast<char *>(id), sizeof(id))) {
              std::ofstream(idfile + ".ack." + std::to_string(rank)) << "read\n"; // rank 0 keeps the file until then
              break;
            }
          }
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
            throw std::runtime_error("rank " + std::to_string(rank) + " timed out after " + std::to_string((int)limit) +
                                     " s waiting for rank 0's communicator id in " + idfile +
                                     (ec ? " (no such file)" : " (only a stale file from an earlier launch)"));
          std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
      }
      check(ctx, aleppo_comm_init(ctx, id));
      if (rank == 0) { // remove the file once every rank has acknowledged reading it (bounded: a rank that died before
                       // it read the id has already failed the collective init above, or will fail the first all-reduce)
        const auto t0 = std::chrono::steady_clock::now();
        std::error_code ec;
        for (int r = 1; r < world; ++r) {
          const std::string ack = idfile + ".ack." + std::to_string(r);
          while (!std::filesystem::exists(ack, ec) && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(300))
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
          std::filesystem::remove(ack, ec);
        }
        std::filesystem::remove(idfile, ec);
      }
      std::cout << "rank " << rank << " of " << world << ": environments [" << env0 << ", " << env0 + E << ")" << std::endl;
    }
    {
      const std::vector<float> p = init_params(cfg.hidden_size, A, cfg.deterministic ? 42 : (uint64_t)start_time);
      size_t n = 0;
      check(ctx, aleppo_param_count(ctx, &n));
      if (n != p.size())
        throw std::runtime_error("parameter count mismatch");
      check(ctx, aleppo_load_params(ctx, p.data(), p.size()));
    }
    EventWriter logger(rank == 0 ? log_path : log_path + ".rank" + std::to_string(rank)); // rank 0's file is THE log
    if (cfg.cuda_graph) // the reference's `cuda_graph: true`: replay the update loop as a captured graph
      check(ctx, aleppo_set_option(ctx, ALEPPO_OPT_UPDATE_GRAPH, 1));
    if (prof.on())
      check(ctx, aleppo_profile_enable(ctx, 1));
    logger.add_hparams( // get_parameters (train.cc:76-105), same keys
        {{"total_environments", (double)cfg.total_environments}, {"hidden_size", (double)cfg.hidden_size},
         {"action_size", (double)cfg.action_size}, {"horizon", (double)cfg.horizon}, {"max_steps", (double)cfg.max_steps},
         {"frame_stack", (double)cfg.frame_stack}, {"learning_rate", cfg.learning_rate}, {"clip_param", cfg.clip_param},
         {"value_loss_coef", cfg.value_loss_coef}, {"entropy_coef", cfg.entropy_coef},
         {"num_epochs", (double)cfg.num_epochs}, {"mini_batch_size", (double)cfg.mini_batch_size},
         {"num_mini_batches", (double)cfg.num_mini_batches}, {"gae_discount", cfg.gae_discount},
         {"gae_lambda", cfg.gae_lambda}, {"max_gradient_norm", cfg.max_gradient_norm},
         {"num_rollouts", (double)cfg.num_rollouts}, {"num_workers", (double)cfg.num_workers},
         {"worker_batch_size", (double)cfg.worker_batch_size}, {"frame_skip", (double)cfg.frame_skip},
         {"max_return", cfg.max_return}},
        {{"record_observation", cfg.record_observation}, {"record_video", cfg.record_video},
         {"cuda_graph", cfg.cuda_graph}, {"deterministic", cfg.deterministic}},
        group, (double)start_time * 1e-9);

    // ---- Rollout host half (src/ai/rollout.cc)
    std::vector<SyntheticAtari> envs;
    for (size_t i = 0; i < E; ++i)
      envs.emplace_back(env0 + i + 0 /*seed arg of train.cc:380*/, cfg.max_steps, cfg.max_return, A, cfg.device_preprocess);
    // the workers' frame buffer: page-locked + GPU-mapped, read in place by the ingest kernel (see the file header)
    uint8_t *frames = nullptr;
    const size_t fbytes = SyntheticAtari::frame_bytes(cfg.device_preprocess); // per environment
    check(ctx, aleppo_host_alloc(ctx, E * fbytes, reinterpret_cast<void **>(&frames)));
    if (cfg.device_preprocess) { // ALE's palette -> gray table would go here; the synthetic palette is its own gray value
      uint8_t lut[256];
      for (int i = 0; i < 256; ++i)
        lut[i] = (uint8_t)i;
      check(ctx, aleppo_set_gray_lut(ctx, lut));
    }
    // episode-start flags at slot entry, where the armed ingest kernel reads them (mapped like the frames)
    uint8_t *start_mapped = nullptr;
    check(ctx, aleppo_host_alloc(ctx, E, reinterpret_cast<void **>(&start_mapped)));
    const bool slot_ahead = cfg.slot_ahead && !prof.on(); // (per-kernel device profiling brackets every launch)
    std::vector<uint8_t> start_cpu(E, 1), term(E, 0), trunc(E, 0), game_over(E, 0);
    std::vector<float> rewards(E, 0.f), ep_ret(E, 0.f), game_ret(E, 0.f);
    std::vector<size_t> ep_len(E, 0), game_len(E, 0);
    std::vector<StepOut> results(E);
    const int64_t *actions = nullptr;
    size_t total_steps = 0, episodes = 0;
    std::cout << "Creating " << cfg.num_workers << " worker threads." << std::endl;
    WorkerPool pool(cfg.num_workers, [&](size_t i) { // Rollout::step (rollout.cc:299-328)
      if (start_cpu[i]) {
        envs[i].reset(&frames[i * fbytes]);
        results[i] = StepOut{};
      } else {
        const int64_t a = actions[i];
        if (a < 0 || (size_t)a >= A)
          throw std::out_of_range("Action index out of range for environment " + std::to_string(i));
        results[i] = envs[i].step((int)a, &frames[i * fbytes]);
      }
    });
    struct Log {
      std::vector<float> episode_returns, game_returns;
      std::vector<size_t> episode_lengths, game_lengths;
    };
    auto rollout = [&]() {
      Log log;
      for (size_t t = 0; t < T; ++t) {
        {
          Profile::Span sp(&prof, "aleppo_act");
          check(ctx, aleppo_act(ctx, nullptr, &actions));
        }
        std::vector<uint8_t> start_at_entry = start_cpu;
        const int fkind = cfg.device_preprocess ? ALEPPO_FRAMES_RAW_PAIR : ALEPPO_FRAMES_84;
        if (slot_ahead) { // slot t + 1's kernels go onto the stream now; they start when release_step lifts the wait
          std::memcpy(start_mapped, start_at_entry.data(), E);
          check(ctx, aleppo_arm_step(ctx, frames, fkind, start_mapped, nullptr));
        }
        {
          Profile::Span sp(&prof, "step_all (emulator threads)");
          pool.run_all(E);
        }
        for (size_t i = 0; i < E; ++i) {
          if (!start_cpu[i]) { // rollout.cc:214-226 (start slots keep the stale reward)
            rewards[i] = results[i].reward;
            term[i] = results[i].terminated;
            trunc[i] = results[i].truncated;
            game_over[i] = results[i].game_over;
            ep_ret[i] += results[i].reward;
            ep_len[i]++;
            game_ret[i] += results[i].reward;
            game_len[i]++;
            total_steps++;
          }
        }
        if (slot_ahead) {
          check(ctx, aleppo_release_step(ctx, rewards.data(), term.data(), trunc.data()));
        } else {
          Profile::Span sp(&prof, "aleppo_step");
          check(ctx, aleppo_step(ctx, frames, fkind, ALEPPO_HOST_MAPPED, rewards.data(), term.data(), trunc.data(),
                                 start_at_entry.data()));
        }
        for (size_t i = 0; i < E; ++i) { // rollout.cc:239-265
          if (results[i].terminated || results[i].truncated) {
            start_cpu[i] = 1;
            term[i] = trunc[i] = 0;
            episodes++;
            log.episode_returns.push_back(ep_ret[i]);
            log.episode_lengths.push_back(ep_len[i]);
            ep_ret[i] = 0;
            ep_len[i] = 0;
            if (game_over[i]) {
              log.game_returns.push_back(game_ret[i]);
              log.game_lengths.push_back(game_len[i]);
              game_ret[i] = 0;
              game_len[i] = 0;
            }
          } else if (start_cpu[i]) {
            start_cpu[i] = 0;
          }
        }
      }
      {
        Profile::Span sp(&prof, "aleppo_finish_rollout");
        check(ctx, aleppo_finish_rollout(ctx, nullptr));
      }
      return log;
    };

    rollout(); // the warm rollout before the loop (train.cc:391-396): collected, never trained on
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<aleppo_minibatch_metrics> m((size_t)cfg.num_epochs * (size_t)cfg.num_mini_batches);
    for (size_t r = 0; r < cfg.num_rollouts; ++r) {
      std::cout << "Rollout " << r + 1 << " of " << cfg.num_rollouts << std::endl;
      const double lr = cfg.learning_rate * (1.0 - r / static_cast<double>(cfg.num_rollouts)); // train.cc:424-428
      const Log log = rollout();
      {
        Profile::Span sp(&prof, "aleppo_train");
        check(ctx, aleppo_train(ctx, lr, (int)cfg.num_epochs, (int)cfg.num_mini_batches, m.data()));
      }
      Profile::Span sp_log(&prof, "log_data");
      // log_data (train.cc:163-210): x axis = non-reset env steps
      const int64_t step = (int64_t)total_steps;
      if (!log.episode_returns.empty()) {
        logger.add_scalar("mean_episode_return", step, meanf(log.episode_returns));
        logger.add_scalar("mean_episode_length", step, meanf(log.episode_lengths));
        logger.add_histogram("episode_returns", step, log.episode_returns);
        logger.add_histogram("episode_lengths", step,
                             std::vector<float>(log.episode_lengths.begin(), log.episode_lengths.end()));
        if (!log.game_returns.empty()) {
          logger.add_scalar("mean_game_return", step, meanf(log.game_returns));
          logger.add_scalar("mean_game_length", step, meanf(log.game_lengths));
          logger.add_histogram("game_returns", step, log.game_returns);
          logger.add_histogram("game_lengths", step, std::vector<float>(log.game_lengths.begin(), log.game_lengths.end()));
        }
      }
      auto avg = [&](float aleppo_minibatch_metrics::*f) {
        double s = 0;
        for (auto &x : m)
          s += x.*f;
        return (float)(s / (double)m.size());
      };
      logger.add_scalar("mean_clipped_gradient", step, avg(&aleppo_minibatch_metrics::grad_norm));
      logger.add_scalar("mean_loss", step, avg(&aleppo_minibatch_metrics::loss));
      logger.add_scalar("mean_clipped_loss", step, avg(&aleppo_minibatch_metrics::clipped_loss));
      logger.add_scalar("mean_value_loss", step, avg(&aleppo_minibatch_metrics::value_loss));
      logger.add_scalar("mean_entropy", step, avg(&aleppo_minibatch_metrics::entropy));
      logger.add_scalar("mean_ratio", step, avg(&aleppo_minibatch_metrics::ratio));
      logger.add_scalar("learning_rate", step, (float)lr);
      {
        std::vector<float> gn;
        for (auto &x : m)
          gn.push_back(x.grad_norm);
        if (gn.size() > 1)
          logger.add_histogram("clipped_gradients", step, gn);
      }
      { // the per-sample histograms of log_data (train.cc:190-207): mask-selected values of the [epochs, M, B] planes
        const size_t N = E * T, per = (size_t)cfg.num_epochs * N;
        std::vector<uint8_t> masks(N);
        std::vector<float> plane(per), adv(N), ret(N), sel;
        check(ctx, aleppo_read_batch(ctx, ALEPPO_F_MASKS, masks.data(), N));
        auto gather = [&](const std::vector<float> &x, size_t reps) { // gather(t, masks): unmasked entries, every epoch
          sel.clear();
          for (size_t r = 0; r < reps; ++r)
            for (size_t i = 0; i < N; ++i)
              if (masks[i])
                sel.push_back(x[r * N + i]);
          return sel;
        };
        const std::pair<int, const char *> fields[5] = {{ALEPPO_M_TOTAL_LOSSES, "losses"},
                                                        {ALEPPO_M_CLIPPED_LOSSES, "clipped_losses"},
                                                        {ALEPPO_M_VALUE_LOSSES, "value_losses"},
                                                        {ALEPPO_M_ENTROPIES, "entropies"},
                                                        {ALEPPO_M_RATIO, "ratios"}};
        for (auto &fd : fields) {
          check(ctx, aleppo_read_train_metric(ctx, fd.first, plane.data(), per));
          logger.add_histogram(fd.second, step, gather(plane, (size_t)cfg.num_epochs));
        }
        check(ctx, aleppo_read_batch(ctx, ALEPPO_F_ADVANTAGES, adv.data(), N * 4));
        check(ctx, aleppo_read_batch(ctx, ALEPPO_F_RETURNS, ret.data(), N * 4));
        logger.add_histogram("advantages", step, gather(adv, 1));
        logger.add_histogram("returns", step, gather(ret, 1));
      }
      logger.flush();
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    size_t pending_starts = 0; // environments whose next slot is an episode-start slot
    for (size_t i = 0; i < E; ++i)
      pending_starts += start_cpu[i];
    // total_steps counts only non-start slots (rollout.cc:225,266): slots = steps + start slots, and the start slots are
    // the E initial ones plus one per finished episode, minus those still pending
    std::cout << "steps " << total_steps << " episodes " << episodes << " pending_starts " << pending_starts << " slots "
              << (cfg.num_rollouts + 1) * E * T << " env-steps/s " << (double)(cfg.num_rollouts * E * T) / secs << std::endl;
    if (prof.on()) {
      prof.device_summary(ctx);
      prof.save();
    }
    if (const char *dump = std::getenv("ALEPPO_TRAINER_DUMP_FINAL")) { // test hook: the trained parameters
      size_t n = 0;
      check(ctx, aleppo_param_count(ctx, &n));
      std::vector<float> p(n);
      check(ctx, aleppo_export_params(ctx, p.data(), n));
      std::ofstream f(dump, std::ios::binary);
      f.write(reinterpret_cast<const char *>(p.data()), (std::streamsize)(n * sizeof(float)));
    }
    check(ctx, aleppo_host_free(ctx, frames));
    check(ctx, aleppo_host_free(ctx, start_mapped));
    aleppo_destroy(ctx);
    std::cout << "Success" << std::endl;
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
}
