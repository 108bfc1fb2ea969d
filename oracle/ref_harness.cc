// TEST INFRASTRUCTURE (oracle/_ref): harness around the UNMODIFIED reference sources.
//
// This file is ours; it is compiled together with the reference's own
//   src/ai/gae.cc, src/ai/buffer.cc, src/ai/ppo/losses.cc, src/ai/ppo/train.cc (+train.h)
// where they lie under /root/reference (recipe: oracle/build_ref.sh, outputs only into
// oracle/_ref/).  Nothing from the reference is copied into this repository.
//
// Two jobs:
//   ref_harness gen <outdir>     run the reference's functions on closed-form hash-fill
//                                inputs (oracle/hashfill.h) and dump the OUTPUTS as .npy
//                                (packed into tests/golden/ref_golden.npz by
//                                oracle/make_golden.py).
//   ref_harness bench E T H A epochs M iters threads
//                                time the reference's CPU-libtorch hot path (Buffer::add /
//                                Buffer::get+gae / ppo::train::train + the action-selector
//                                closure and update_observations restated in libtorch calls)
//                                -> one JSON line.  Used as bench.py's cpu_baseline
//                                ("kind":"reference").
//
// Parts of the reference that cannot be compiled here (src/bin/train.cc needs ALE, yaml-cpp,
// tensorboard_logger; src/ai/rollout.cc needs ALE; src/ai/vision.cc needs stb) are restated
// below IN LIBTORCH CALLS so the arithmetic is still libtorch's:
//   RefNet            <- NetworkImpl            src/bin/train.cc:230-270
//   select_actions    <- action selector lambda src/bin/train.cc:367-379
//   prepare_batch     <- prepare_batch          src/bin/train.cc:272-283
//   update_observations <- Rollout::update_observations  src/ai/rollout.cc:184-196
//   area_resize / luma  <- ai::vision::*        src/ai/vision.cc:8-32, :51, :71-84
#include "ai/buffer.h"
#include "ai/gae.h"
#include "ai/ppo/losses.h"
#include "ai/ppo/train.h"
#include "hashfill.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <torch/torch.h>
#include <vector>

// ---------------------------------------------------------------- npy writer
static void write_npy(const std::string &path, const torch::Tensor &t_in) {
  auto t = t_in.detach().contiguous().cpu();
  std::string descr;
  if (t.dtype() == torch::kFloat32)
    descr = "<f4";
  else if (t.dtype() == torch::kFloat64)
    descr = "<f8";
  else if (t.dtype() == torch::kInt64)
    descr = "<i8";
  else if (t.dtype() == torch::kUInt8)
    descr = "|u1";
  else if (t.dtype() == torch::kBool) {
    t = t.to(torch::kUInt8);
    descr = "|u1";
  } else
    throw std::runtime_error("write_npy: dtype");
  std::string shape = "(";
  for (int64_t i = 0; i < t.dim(); ++i)
    shape += std::to_string(t.size(i)) + ",";
  shape += ")";
  std::string hdr = "{'descr': '" + descr + "', 'fortran_order': False, 'shape': " + shape + ", }";
  size_t total = 10 + hdr.size() + 1;
  size_t pad = (64 - total % 64) % 64;
  hdr += std::string(pad, ' ');
  hdr += "\n";
  std::ofstream f(path, std::ios::binary);
  const char magic[] = "\x93NUMPY\x01\x00";
  f.write(magic, 8);
  uint16_t hl = (uint16_t)hdr.size();
  f.write(reinterpret_cast<const char *>(&hl), 2);
  f.write(hdr.data(), hdr.size());
  f.write(reinterpret_cast<const char *>(t.data_ptr()), t.numel() * t.element_size());
}

// ---------------------------------------------------------------- hash fills as tensors
static torch::Tensor fill_range(uint32_t seed, std::vector<int64_t> shape, float lo, float hi) {
  auto t = torch::empty(shape, torch::kFloat32);
  float *p = t.data_ptr<float>();
  for (int64_t i = 0; i < t.numel(); ++i)
    p[i] = hf_range(seed, (uint32_t)i, lo, hi);
  return t;
}
static torch::Tensor fill_bytes(uint32_t seed, std::vector<int64_t> shape) {
  auto t = torch::empty(shape, torch::kUInt8);
  uint8_t *p = t.data_ptr<uint8_t>();
  for (int64_t i = 0; i < t.numel(); ++i)
    p[i] = hf_byte(seed, (uint32_t)i);
  return t;
}

// ---------------------------------------------------------------- network restated in libtorch
// Topology / forward of NetworkImpl (src/bin/train.cc:232-244, :255-265): conv(4->32,k8,s4) relu
// conv(32->64,k4,s2) relu conv(64->64,k3,s1) relu flatten linear(3136->H) [no relu]
// -> {linear(H->A), linear(H->1).squeeze(-1)};  uint8 -> f32 / 255 inside forward, no grad.
struct RefNetImpl : torch::nn::Module {
  std::vector<torch::Tensor> p; // libtorch parameters() order of the reference module tree
  RefNetImpl(int64_t H, int64_t A) {
    auto reg = [&](const char *n, std::vector<int64_t> s) {
      p.push_back(register_parameter(n, torch::zeros(s)));
    };
    reg("c1w", {32, 4, 8, 8});
    reg("c1b", {32});
    reg("c2w", {64, 32, 4, 4});
    reg("c2b", {64});
    reg("c3w", {64, 64, 3, 3});
    reg("c3b", {64});
    reg("fcw", {H, 3136});
    reg("fcb", {H});
    reg("aw", {A, H});
    reg("ab", {A});
    reg("vw", {1, H});
    reg("vb", {1});
  }
  struct Out {
    torch::Tensor logits, value;
  };
  Out forward(torch::Tensor x) {
    {
      torch::NoGradGuard ng;
      x = x.to(torch::kFloat32);
      x.divide_(255.0);
    }
    x = torch::relu(torch::conv2d(x, p[0], p[1], 4));
    x = torch::relu(torch::conv2d(x, p[2], p[3], 2));
    x = torch::relu(torch::conv2d(x, p[4], p[5], 1));
    x = x.flatten(1);
    x = torch::linear(x, p[6], p[7]);
    auto logits = torch::linear(x, p[8], p[9]);
    auto value = torch::linear(x, p[10], p[11]).squeeze(-1);
    return {logits, value};
  }
};
using RefNet = std::shared_ptr<RefNetImpl>;

// closed-form parameter fill shared with tests/hashfill.py: tensor k gets
// U(-b,b), b = sqrt(6/fan_in) for weights, 0.05 for biases; seed = base + k
static const int64_t FAN_IN[12] = {256, 0, 512, 0, 576, 0, 3136, 0, -1, 0, -1, 0};
static void fill_params(RefNet &net, uint32_t seed_base, int64_t H) {
  torch::NoGradGuard ng;
  for (int k = 0; k < 12; ++k) {
    auto &t = net->p[k];
    float b;
    if (k % 2 == 1)
      b = 0.05f;
    else {
      int64_t fi = FAN_IN[k] < 0 ? H : FAN_IN[k];
      b = std::sqrt(6.0f / (float)fi);
    }
    float *d = t.data_ptr<float>();
    for (int64_t i = 0; i < t.numel(); ++i)
      d[i] = hf_range(seed_base + k, (uint32_t)i, -b, b);
  }
}

// prepare_batch (src/bin/train.cc:272-283)
static ai::ppo::train::Batch prepare_batch(ai::buffer::Batch &b) {
  auto lp = ai::ppo::losses::normalize_logits(b.logits.view({-1, b.logits.size(2)}));
  return {b.observations.flatten(0, 1), b.actions.ravel(), lp, b.advantages.ravel(), b.returns.ravel(),
          b.masks.ravel()};
}

// update_observations (src/ai/rollout.cc:184-196) on obs [E,S,84,84] with per-env frames
static void update_observations(torch::Tensor &obs, const std::vector<torch::Tensor> &frames,
                                const std::vector<bool> &start) {
  using torch::indexing::Slice;
  int64_t S = obs.size(1);
  for (int64_t f = S - 1; f > 0; --f)
    obs.index_put_({Slice(), f}, obs.index({Slice(), f - 1}));
  for (size_t i = 0; i < frames.size(); ++i)
    if (start[i])
      obs.select(0, (int64_t)i).copy_(frames[i], true);
  obs.index_put_({Slice(), 0}, torch::stack(frames, 0));
}

static torch::Tensor plane_checksums(const torch::Tensor &obs) { // [E,S,84,84] u8 -> [E,S,2] i64
  auto o = obs.to(torch::kInt64).flatten(2);
  auto w = torch::arange(1, o.size(2) + 1, torch::kInt64);
  return torch::stack({o.sum(-1), (o * w).sum(-1)}, -1);
}

static torch::Tensor sample_entries(const torch::Tensor &t, uint32_t seed, int n) {
  auto f = t.detach().flatten();
  auto out = torch::empty({n}, torch::kFloat32);
  for (int j = 0; j < n; ++j)
    out[j] = f[(int64_t)(hf_u32(seed, (uint32_t)j) % (uint32_t)f.numel())];
  return out;
}

// ---------------------------------------------------------------- gen
static void gen(const std::string &out) {
  torch::manual_seed(42);
  auto W = [&](const std::string &n, const torch::Tensor &t) { write_npy(out + "/" + n + ".npy", t); };

  // ---- KAT: the six gae-test.cc inputs (test/ai/gae-test.cc:6,41,80,114,148,205), outputs of the
  // real ai::gae::gae.  Inputs are re-typed from the NUMBERS in that file by tests/test_oracle.py.
  {
    struct Case {
      int E, T;
      std::vector<float> r, v, nv;
      std::vector<int> term, trunc, start;
    };
    std::vector<Case> cs = {
        {1, 3, {1, 1, 1}, {.5, .5, .5}, {.5}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}},
        {1, 3, {1, 1, 1}, {.5, .5, .5}, {0}, {0, 0, 1}, {0, 0, 0}, {0, 0, 0}},
        {1, 3, {1, 1, 1}, {.5, .5, .5}, {0}, {0, 0, 0}, {0, 0, 1}, {0, 0, 0}},
        {1, 3, {1, 1, 1}, {.5, .5, .5}, {.5}, {0, 0, 0}, {0, 0, 0}, {0, 1, 0}},
        {2, 3, {1, 1, 1, .5, .5, .5}, {.5, .5, .5, .3, .3, .3}, {.5, .3}, {0, 0, 0, 0, 0, 1}, {0, 0, 0, 0, 0, 0},
         {0, 1, 0, 0, 0, 0}},
        {1, 5, {1, 1, 1, 1, 1}, {.5, .5, .5, .5, .5}, {.5}, {0, 0, 1, 0, 0}, {0, 0, 0, 0, 1}, {0, 0, 0, 1, 0}},
    };
    int k = 0;
    for (auto &c : cs) {
      auto r = torch::tensor(c.r).view({c.E, c.T});
      auto v = torch::tensor(c.v).view({c.E, c.T});
      auto nv = torch::tensor(c.nv);
      auto mk = [&](std::vector<int> &f) { return torch::tensor(f).view({c.E, c.T}).to(torch::kBool); };
      auto adv = torch::zeros({c.E, c.T});
      ai::gae::gae(adv, r, v, nv, mk(c.term), mk(c.trunc), mk(c.start), 0.99f, 0.95f);
      W("kat" + std::to_string(k++) + "_adv", adv);
    }
  }

  // ---- G1: reward clamp + GAE + returns through the real Buffer::add / Buffer::get, [128,128]
  {
    const int64_t E = 128, T = 128;
    ai::buffer::Buffer buf(E, T, {1}, 4, torch::kCPU);
    auto nv = torch::empty({E});
    std::vector<bool> prev_end(E, false);
    for (int64_t t = 0; t < T; ++t) {
      auto r = torch::zeros({E}), v = torch::zeros({E});
      auto term = torch::zeros({E}, torch::kBool), trunc = torch::zeros({E}, torch::kBool),
           start = torch::zeros({E}, torch::kBool);
      for (int64_t e = 0; e < E; ++e) {
        uint32_t idx = (uint32_t)(e * T + t);
        r[e] = hf_unit(101, idx) < 0.3f ? hf_range(102, idx, -3.f, 3.f) : 0.f;
        v[e] = hf_range(103, idx, -1.f, 1.f);
        bool st = (t == 0 && hf_unit(106, (uint32_t)e) < 0.25f) || prev_end[e];
        bool te = false, tr = false;
        if (!st) {
          float u = hf_unit(105, idx);
          te = u < 0.03f;
          tr = !te && u < 0.04f;
        }
        prev_end[e] = te || tr;
        start[e] = st;
        term[e] = te;
        trunc[e] = tr;
      }
      buf.add(torch::zeros({E, 1}, torch::kUInt8), torch::zeros({E}, torch::kInt64), r, term, trunc, start,
              torch::zeros({E, 4}), v);
    }
    for (int64_t e = 0; e < E; ++e)
      nv[e] = hf_range(104, (uint32_t)e, -1.f, 1.f);
    auto b = buf.get(nv, 0.99f, 0.95f);
    W("g1_adv", b.advantages);
    W("g1_returns", b.returns);
    W("g1_rewards_clamped", b.rewards);
    W("g1_masks", b.masks);
  }

  // ---- G2: ai::ppo::losses::compute + autograd, B=256, A in {4,6}
  for (int64_t A : {4, 6}) {
    const int64_t B = 256;
    auto logits = fill_range(201 + A, {B, A}, -2.f, 2.f).requires_grad_(true);
    auto old_logits = logits.detach() + fill_range(203 + A, {B, A}, -0.6f, 0.6f);
    auto old_lp = ai::ppo::losses::normalize_logits(old_logits);
    auto actions = torch::empty({B}, torch::kInt64);
    auto masks = torch::empty({B}, torch::kBool);
    for (int64_t i = 0; i < B; ++i) {
      actions[i] = (int64_t)(hf_u32(205 + A, (uint32_t)i) % (uint32_t)A);
      masks[i] = hf_unit(209, (uint32_t)i) >= 0.05f;
    }
    auto adv = fill_range(206, {B}, -2.f, 2.f);
    auto values = fill_range(207, {B}, -1.f, 1.f).requires_grad_(true);
    auto returns = fill_range(208, {B}, -1.5f, 1.5f);
    auto lp = ai::ppo::losses::normalize_logits(logits);
    auto m = ai::ppo::losses::compute(lp, old_lp, actions, adv, values, returns, masks, 0.1f, 0.5f, 0.01f);
    m.loss.backward();
    std::string s = "g2_A" + std::to_string(A) + "_";
    W(s + "old_logp", old_lp);
    W(s + "loss", m.loss.reshape({1}));
    W(s + "clipped", m.clipped_losses);
    W(s + "value_losses", m.value_losses);
    W(s + "entropies", m.entropies);
    W(s + "total_losses", m.total_losses);
    W(s + "ratio", m.ratio);
    W(s + "dlogits", logits.grad());
    W(s + "dvalues", values.grad());
  }

  // ---- G3: network forward, N=8, H in {32,512}, A in {4,6}
  {
    auto obs = fill_bytes(301, {8, 4, 84, 84});
    for (int64_t H : {32, 512})
      for (int64_t A : {4, 6}) {
        RefNet net = std::make_shared<RefNetImpl>(H, A);
        fill_params(net, 310, H);
        torch::NoGradGuard ng;
        net->eval();
        auto o = net->forward(obs);
        std::string s = "g3_H" + std::to_string(H) + "_A" + std::to_string(A) + "_";
        W(s + "logits", o.logits);
        W(s + "values", o.value);
      }
  }

  // ---- G4: the real ai::ppo::train::train, H=32, A=4, N=64
  {
    const int64_t H = 32, A = 4, N = 64;
    auto obs = fill_bytes(401, {N, 4, 84, 84});
    auto actions = torch::empty({N}, torch::kInt64);
    auto masks = torch::empty({N}, torch::kBool);
    for (int64_t i = 0; i < N; ++i) {
      actions[i] = (int64_t)(hf_u32(402, (uint32_t)i) % (uint32_t)A);
      masks[i] = hf_unit(406, (uint32_t)i) >= 0.1f;
    }
    auto old_lp = ai::ppo::losses::normalize_logits(fill_range(403, {N, A}, -1.f, 1.f));
    auto adv = fill_range(404, {N}, -1.f, 1.f);
    auto ret = fill_range(405, {N}, -1.f, 1.f);
    W("g4_old_logp", old_lp);
    struct Cfg {
      const char *name;
      size_t epochs, M;
    };
    for (Cfg c : {Cfg{"a", 1, 1}, Cfg{"b", 2, 4}}) {
      RefNet net = std::make_shared<RefNetImpl>(H, A);
      fill_params(net, 410, H);
      torch::optim::Adam opt(net->parameters(), torch::optim::AdamOptions(2.5e-4).eps(1e-5));
      ai::ppo::train::Batch batch{obs, actions, old_lp, adv, ret, masks};
      ai::ppo::train::Metrics metrics((int64_t)c.epochs, (int64_t)c.M, N / (int64_t)c.M, torch::kCPU);
      auto indices = torch::empty({N}, torch::kInt64);
      ai::ppo::train::Hyperparameters hp{0.1f, 0.5f, 0.01f, 0.5f};
      ai::ppo::train::train(net, opt, metrics, indices, batch, c.epochs, c.M, hp);
      std::string s = std::string("g4") + c.name + "_";
      using torch::indexing::Slice;
      W(s + "loss", metrics.loss.index({Slice(), Slice(), 0}));
      W(s + "grad_norm", metrics.clipped_gradients);
      W(s + "total_losses", metrics.total_losses);
      W(s + "ratio", metrics.ratio);
      W(s + "entropies", metrics.entropies);
      W(s + "value_losses", metrics.value_losses);
      W(s + "clipped", metrics.clipped_losses);
      std::vector<torch::Tensor> sums, samples, gsums, gsamples;
      for (int k = 0; k < 12; ++k) {
        auto t = net->p[k].detach().to(torch::kFloat64).flatten();
        sums.push_back(torch::stack({t.sum(), (t * t).sum()}));
        samples.push_back(sample_entries(net->p[k], 499, 64));
        auto g = net->p[k].grad().to(torch::kFloat64).flatten();
        gsums.push_back(torch::stack({g.sum(), (g * g).sum()}));
        gsamples.push_back(sample_entries(net->p[k].grad(), 499, 64));
      }
      W(s + "param_sums", torch::stack(sums));
      W(s + "param_samples", torch::stack(samples));
      W(s + "grad_sums", torch::stack(gsums));       // clipped grads of the LAST minibatch
      W(s + "grad_samples", torch::stack(gsamples)); //   (what clip_grad_norm_ left in .grad)
      for (int k : {0, 1, 3, 5, 7, 8, 9, 10, 11})
        W(s + "param" + std::to_string(k), net->p[k]);
    }
  }

  // ---- G5: sampling, torch::multinomial(probs,1,true) vs its exponential noise (src/bin/train.cc:374-375)
  for (auto [E, A] : {std::pair<int64_t, int64_t>{128, 4}, {512, 6}}) {
    auto probs = torch::softmax(fill_range(501 + A, {E, A}, -3.f, 3.f), -1);
    torch::manual_seed(7 + A);
    auto q = torch::empty_like(probs).exponential_(1);
    torch::manual_seed(7 + A);
    auto actions = torch::multinomial(probs, 1, true).ravel();
    std::string s = "g5_A" + std::to_string(A) + "_";
    W(s + "probs", probs);
    W(s + "q", q);
    W(s + "actions", actions);
  }

  // ---- G6: frame stack, 6 steps on [4,4,84,84] with start events
  {
    const int64_t E = 4;
    auto obs = torch::zeros({E, 4, 84, 84}, torch::kUInt8);
    std::vector<torch::Tensor> sums;
    for (int step = 0; step < 6; ++step) {
      std::vector<torch::Tensor> frames;
      std::vector<bool> start(E, false);
      for (int64_t e = 0; e < E; ++e) {
        auto f = torch::empty({84, 84}, torch::kUInt8);
        uint8_t *p = f.data_ptr<uint8_t>();
        for (int i = 0; i < 7056; ++i)
          p[i] = hf_byte(601 + step, (uint32_t)(e * 7056 + i));
        frames.push_back(f);
        start[e] = step == 0 || (step == 3 && e == 1) || (step == 4 && e == 2);
      }
      update_observations(obs, frames, start);
      sums.push_back(plane_checksums(obs));
    }
    W("g6_checksums", torch::stack(sums));
    W("g6_final_obs_env1", obs.select(0, 1));
  }

  // ---- G7: area resize (src/ai/vision.cc:8-10,22-32) and luma (:51,:71-84)
  {
    auto opts = torch::nn::functional::InterpolateFuncOptions().size(std::vector<int64_t>({84, 84})).mode(torch::kArea);
    auto in = fill_bytes(701, {1, 2, 210, 160}).to(torch::kFloat32);
    auto o = torch::nn::functional::interpolate(in.flatten(0, 1).unsqueeze(1), opts).squeeze(1).unflatten(0, {1, 2});
    W("g7_area", o);
    auto rgb = fill_bytes(702, {1, 1, 3, 84, 84}).to(torch::kFloat32);
    auto w = torch::tensor({0.2125, 0.7154, 0.0721}).to(torch::kFloat32);
    W("g7_luma", torch::matmul(rgb.permute({0, 1, 3, 4, 2}), w));
  }
  std::cout << "gen ok -> " << out << std::endl;
}

// ---------------------------------------------------------------- bench (cpu_baseline "reference")
static int bench(int64_t E, int64_t T, int64_t H, int64_t A, size_t epochs, size_t M, int iters, int threads) {
  torch::set_num_threads(threads);
  torch::manual_seed(42);
  RefNet net = std::make_shared<RefNetImpl>(H, A);
  fill_params(net, 310, H);
  torch::optim::Adam opt(net->parameters(), torch::optim::AdamOptions(2.5e-4).eps(1e-5));
  ai::buffer::Buffer buf(E, T, {4, 84, 84}, A, torch::kCPU);
  auto obs = torch::zeros({E, 4, 84, 84}, torch::kUInt8);
  auto rewards = torch::zeros({E});
  auto term = torch::zeros({E}, torch::kBool), trunc = torch::zeros({E}, torch::kBool),
       start = torch::ones({E}, torch::kBool);
  std::vector<std::vector<uint8_t>> host_frames(E, std::vector<uint8_t>(7056));
  std::vector<torch::Tensor> blobs;
  for (int64_t e = 0; e < E; ++e)
    blobs.push_back(torch::from_blob(host_frames[e].data(), {84, 84}, torch::kUInt8));
  std::vector<bool> start_cpu(E, true);
  ai::ppo::train::Metrics metrics((int64_t)epochs, (int64_t)M, E * T / (int64_t)M, torch::kCPU);
  auto indices = torch::empty({E * T}, torch::kInt64);
  ai::ppo::train::Hyperparameters hp{0.1f, 0.5f, 0.01f, 0.5f};
  auto select = [&](const torch::Tensor &o) {
    net->eval();
    torch::NoGradGuard ng;
    auto out = net->forward(o);
    auto probs = torch::softmax(out.logits, -1);
    auto actions = torch::multinomial(probs, 1, true);
    return std::make_tuple(actions.ravel(), out.logits.reshape({-1, A}), out.value.ravel());
  };
  double total_s = 0;
  uint32_t ctr = 0;
  for (int it = 0; it < iters + 1; ++it) { // first iteration is warm-up
    auto t0 = std::chrono::steady_clock::now();
    {
      torch::NoGradGuard ng;
      for (int64_t t = 0; t < T; ++t) {
        auto [actions, logits, values] = select(obs);
        for (int64_t e = 0; e < E; ++e) { // synthetic env step + the per-env scalar writes
          uint32_t idx = ctr++;
          for (int i = 0; i < 7056; i += 64)
            host_frames[e][i] = hf_byte(900, idx + i);
          if (!start_cpu[e]) {
            float u = hf_unit(901, idx);
            rewards[e] = u < 0.05f ? 1.f : 0.f;
            term[e] = hf_unit(902, idx) < 0.005f;
            trunc[e] = false;
          }
        }
        buf.add(obs, actions, rewards, term, trunc, start, logits, values);
        update_observations(obs, blobs, start_cpu);
        for (int64_t e = 0; e < E; ++e) {
          bool ended = term[e].item<bool>() || trunc[e].item<bool>();
          if (ended) {
            start[e] = true;
            term[e] = false;
            trunc[e] = false;
            start_cpu[e] = true;
          } else if (start_cpu[e]) {
            start[e] = false;
            start_cpu[e] = false;
          }
        }
      }
    }
    ai::buffer::Batch b;
    {
      torch::NoGradGuard ng;
      auto last = select(obs);
      b = buf.get(std::get<2>(last), 0.99f, 0.95f);
    }
    auto batch = prepare_batch(b);
    ai::ppo::train::train(net, opt, metrics, indices, batch, epochs, M, hp);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (it > 0)
      total_s += s;
  }
  double sps = (double)(E * T) * iters / total_s;
  std::printf("{\"env_steps_per_s\": %.3f, \"threads\": %d, \"E\": %ld, \"T\": %ld, \"H\": %ld, \"A\": %ld, "
              "\"epochs\": %zu, \"M\": %zu, \"iters\": %d, \"seconds\": %.3f}\n",
              sps, threads, (long)E, (long)T, (long)H, (long)A, epochs, M, iters, total_s);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 3 && std::string(argv[1]) == "gen") {
    gen(argv[2]);
    return 0;
  }
  if (argc >= 10 && std::string(argv[1]) == "bench")
    return bench(atol(argv[2]), atol(argv[3]), atol(argv[4]), atol(argv[5]), (size_t)atol(argv[6]),
                 (size_t)atol(argv[7]), atoi(argv[8]), atoi(argv[9]));
  std::fprintf(stderr, "usage: ref_harness gen <outdir> | bench E T H A epochs M iters threads\n");
  return 2;
}
