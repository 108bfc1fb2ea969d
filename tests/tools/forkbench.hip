// What does a cross-stream dependency cost inside the update?  (the dgrad chain on the main stream forks
// three times per minibatch to the weight-gradient stream and joins once.)
//
// One "minibatch" here: main stream M runs busy kernels A -> B -> C, side stream S runs D after A (fork) and C waits for D
// (join); every busy kernel spins for SPIN us on all CUs.  Perfect overlap gives 3 x SPIN per iteration.  Variants:
//   serial  : A, D, B, C on one stream (4 x SPIN + boundaries)                          - no dependency machinery at all
//   events  : hipEventRecord + hipStreamWaitEvent for fork and join                      - what aleppo_train does
//   signal  : a one-wave kernel on the producing stream stores a sequence number into a device word, a one-wave gate
//             kernel on the consuming stream polls it (device memory, sc1 loads)
//   inkern  : the producing busy kernel's LAST workgroup stores the sequence number itself (arrival counter), gate kernel
//             on the consuming stream
// hipcc --offload-arch=gfx950 -O2 tests/tools/forkbench.hip -o build_tools/forkbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// every workgroup spins for `ticks` of the 100 MHz wall clock; optionally the last one to finish publishes `seq`
__global__ __launch_bounds__(256) void busy(unsigned long long ticks, unsigned int *arrive, unsigned long long *flag,
                                            unsigned long long seq, float *sink) {
  const unsigned long long t0 = wall_clock64();
  float x = threadIdx.x;
  while (wall_clock64() - t0 < ticks)
    x = x * 1.0001f + 0.5f;
  if (x == 12345.678f)
    sink[0] = x;
  if (flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int prev = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == gridDim.x - 1) {
        __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
__global__ void signal_k(unsigned long long *flag, unsigned long long seq) {
  if (threadIdx.x == 0)
    __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void gate_k(unsigned long long *flag, unsigned long long seq, unsigned long long timeout_ticks,
                       unsigned long long *report) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > timeout_ticks) { // exit condition: never spin for ever
        __hip_atomic_store(report, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
}

int main(int argc, char **argv) {
  const double spin_us = argc > 1 ? std::atof(argv[1]) : 30.0;
  const int iters = 300, grid = 256;
  const unsigned long long ticks = (unsigned long long)(spin_us * 100.0), tmo = 200000000ull; // 2 s
  hipStream_t M, S;
  CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  hipEvent_t ef, ej;
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  unsigned long long *flags;
  unsigned int *arrive;
  float *sink;
  CK(hipMalloc(reinterpret_cast<void **>(&flags), 4096));
  CK(hipMalloc(reinterpret_cast<void **>(&arrive), 4096));
  CK(hipMalloc(reinterpret_cast<void **>(&sink), 4096));
  CK(hipMemset(flags, 0, 4096));
  CK(hipMemset(arrive, 0, 4096));
  CK(hipDeviceSynchronize());
  unsigned long long *ffork = flags, *fjoin = flags + 64, *rep = flags + 128; // own 512-byte regions
  unsigned int *af = arrive, *aj = arrive + 64;
  auto B = [&](hipStream_t st, unsigned int *a = nullptr, unsigned long long *f = nullptr, unsigned long long seq = 0) {
    hipLaunchKernelGGL(busy, dim3(grid), dim3(256), 0, st, ticks, a, f, seq, sink);
  };
  unsigned long long seq = 0;
  auto run = [&](int variant) {
    for (int it = 0; it < iters; ++it) {
      ++seq;
      if (variant == 0) {
        B(M); B(M); B(M); B(M);
      } else if (variant == 1) {
        B(M);
        CK(hipEventRecord(ef, M));
        CK(hipStreamWaitEvent(S, ef, 0));
        B(S);
        CK(hipEventRecord(ej, S));
        B(M);
        CK(hipStreamWaitEvent(M, ej, 0));
        B(M);
      } else if (variant == 2) {
        B(M);
        hipLaunchKernelGGL(signal_k, dim3(1), dim3(64), 0, M, ffork, seq);
        hipLaunchKernelGGL(gate_k, dim3(1), dim3(64), 0, S, ffork, seq, tmo, rep);
        B(S);
        hipLaunchKernelGGL(signal_k, dim3(1), dim3(64), 0, S, fjoin, seq);
        B(M);
        hipLaunchKernelGGL(gate_k, dim3(1), dim3(64), 0, M, fjoin, seq, tmo, rep);
        B(M);
      } else {
        B(M, af, ffork, seq);
        hipLaunchKernelGGL(gate_k, dim3(1), dim3(64), 0, S, ffork, seq, tmo, rep);
        B(S, aj, fjoin, seq);
        B(M);
        hipLaunchKernelGGL(gate_k, dim3(1), dim3(64), 0, M, fjoin, seq, tmo, rep);
        B(M);
      }
    }
  };
  const char *names[4] = {"serial (one stream)", "events", "signal kernel + gate kernel", "in-kernel signal + gate kernel"};
  std::printf("busy kernels of %.0f us on %d workgroups; ideal overlapped iteration = %.0f us\n", spin_us, grid, 3 * spin_us);
  for (int rep_i = 0; rep_i < 2; ++rep_i)
    for (int v = 0; v < 4; ++v) {
      run(v); // warm-up (also keeps seq monotonic)
      CK(hipStreamSynchronize(M));
      CK(hipStreamSynchronize(S));
      const auto t0 = std::chrono::steady_clock::now();
      run(v);
      CK(hipStreamSynchronize(M));
      CK(hipStreamSynchronize(S));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      unsigned long long r = 0;
      CK(hipMemcpy(&r, rep, 8, hipMemcpyDeviceToHost));
      std::printf("%-34s %8.1f us per iteration%s\n", names[v], us, r ? "  (a gate timed out!)" : "");
      std::fflush(stdout);
    }
  return 0;
}
