// Which HIP runtime calls wait for a stream that is PARKED behind a host-released gate - and do they block the
// owner thread's next enqueue on that stream meanwhile?  ("several contexts in one process".)
//
// Set-up per probe: stream S (non-blocking) holds a one-wave gate kernel spinning on a mapped host word; only the main
// thread writes that word, 1.5 s after the probe started.  Thread F ("foreign") calls the probed runtime function at
// t = 0; thread L ("owner") calls hipLaunchKernelGGL(noop, S) at t = 0.3 s.  Reported: how long each call took.
//   F ~ 1.5 s : the call waits for the parked stream (an implicit device-wide synchronisation)
//   L ~ 1.2 s : ... and while it waits, the owner's enqueue on S blocks too.  An owner that releases the gate only
//               AFTER its enqueues (the slot-ahead hand-off does) would then never release it: a dead-lock.
// hipcc --offload-arch=gfx950 -O2 tests/tools/parkprobe.hip -o build_tools/parkprobe -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void gate(const unsigned long long *go, unsigned long long seq, unsigned long long *report,
                     unsigned long long timeout_ticks) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > timeout_ticks) { // exit condition every wave reaches: never spin for ever
        __hip_atomic_store(report, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
}
__global__ void fresh(int *p) {
  if (threadIdx.x == 1)
    p[1] = 2;
}
__global__ void fresh2(int *p) {
  extern __shared__ int sm[];
  sm[threadIdx.x] = 1;
  if (threadIdx.x == 2)
    p[2] = sm[3];
}
__global__ void noop(int *p) {
  if (p && threadIdx.x == 0)
    *p = 1;
}

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  unsigned long long *go = nullptr;
  CK(hipHostMalloc(reinterpret_cast<void **>(&go), 64, hipHostMallocMapped));
  hipStream_t S, S2;
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&S2, hipStreamNonBlocking));
  int *dscratch = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&dscratch), 1 << 20));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  std::vector<char> hostbuf(1 << 16);
  void *pinned = nullptr;
  CK(hipHostMalloc(&pinned, 1 << 16, hipHostMallocDefault));

  auto new_streams = [&](int n) {
    std::vector<hipStream_t> v(n);
    for (auto &t : v)
      CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
    for (auto &t : v)
      hipLaunchKernelGGL(noop, dim3(1), dim3(64), 0, t, dscratch);
    for (auto &t : v)
      CK(hipStreamSynchronize(t));
    for (auto &t : v)
      CK(hipStreamDestroy(t));
  };
  struct Probe {
    const char *name;
    std::function<void()> fn;
  };
  void *victim = nullptr, *hvictim = nullptr, *extra = nullptr;
  std::vector<void *> leak;
  std::vector<char> bighost(8 << 20);
  void *big = nullptr;
  CK(hipMalloc(&big, 8 << 20));
  std::vector<Probe> probes = {
      {"hipFree", [&] { CK(hipFree(victim)); }},
      {"hipHostFree", [&] { CK(hipHostFree(hvictim)); }},
      {"hipMalloc", [&] { CK(hipMalloc(&extra, 1 << 20)); }},
      {"hipDeviceSynchronize", [&] { CK(hipDeviceSynchronize()); }},
      {"hipStreamSynchronize(null)", [&] { CK(hipStreamSynchronize(nullptr)); }},
      {"hipMemset(null stream)", [&] { CK(hipMemset(dscratch, 0, 4096)); CK(hipStreamSynchronize(nullptr)); }},
      {"hipMemcpy H2D pageable", [&] { CK(hipMemcpy(dscratch, hostbuf.data(), hostbuf.size(), hipMemcpyHostToDevice)); }},
      {"hipMemcpy D2H pageable", [&] { CK(hipMemcpy(hostbuf.data(), dscratch, hostbuf.size(), hipMemcpyDeviceToHost)); }},
      {"hipMemcpyAsync+sync other stream", [&] { CK(hipMemcpyAsync(dscratch, pinned, 1 << 16, hipMemcpyHostToDevice, S2)); CK(hipStreamSynchronize(S2)); }},
      {"kernel+sync other stream", [&] { hipLaunchKernelGGL(noop, dim3(1), dim3(64), 0, S2, dscratch); CK(hipStreamSynchronize(S2)); }},
      {"hipStreamCreate+Destroy", [&] { hipStream_t t; CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking)); CK(hipStreamDestroy(t)); }},
      {"hipEventRecord(other)+Synchronize", [&] { CK(hipEventRecord(ev, S2)); CK(hipEventSynchronize(ev)); }},
      {"hipHostMalloc(default)", [&] { void *q; CK(hipHostMalloc(&q, 1 << 16, hipHostMallocDefault)); leak.push_back(q); }},
      {"hipHostMalloc(mapped)", [&] { void *q; CK(hipHostMalloc(&q, 1 << 16, hipHostMallocMapped)); leak.push_back(q); }},
      {"hipHostGetDevicePointer", [&] { void *d; CK(hipHostGetDevicePointer(&d, pinned, 0)); }},
      {"hipEventCreate+Destroy", [&] { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventDestroy(e)); }},
      {"hipMemsetAsync+sync other stream", [&] { CK(hipMemsetAsync(dscratch, 0, 1 << 16, S2)); CK(hipStreamSynchronize(S2)); }},
      {"MemcpyAsync H2D pageable other+sync", [&] { CK(hipMemcpyAsync(dscratch, hostbuf.data(), hostbuf.size(), hipMemcpyHostToDevice, S2)); CK(hipStreamSynchronize(S2)); }},
      {"MemcpyAsync D2H pageable other+sync", [&] { CK(hipMemcpyAsync(hostbuf.data(), dscratch, hostbuf.size(), hipMemcpyDeviceToHost, S2)); CK(hipStreamSynchronize(S2)); }},
      {"MemcpyAsync H2D 8MB pageable other", [&] { CK(hipMemcpyAsync(big, bighost.data(), bighost.size(), hipMemcpyHostToDevice, S2)); CK(hipStreamSynchronize(S2)); }},
      {"MemcpyAsync D2H 8MB pageable other", [&] { CK(hipMemcpyAsync(bighost.data(), big, bighost.size(), hipMemcpyDeviceToHost, S2)); CK(hipStreamSynchronize(S2)); }},
      {"hipStreamQuery(parked)", [&] { (void)hipStreamQuery(S); }},
      {"hipGetDeviceProperties+SetDevice", [&] { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); CK(hipSetDevice(0)); }},
      {"first launch of a new kernel (other)", [&] { hipLaunchKernelGGL(fresh, dim3(1), dim3(64), 0, S2, dscratch); CK(hipStreamSynchronize(S2)); }},
      {"hipFuncSetAttribute", [&] { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fresh2), hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); }},
      {"hipMalloc 1 GB", [&] { CK(hipMalloc(&extra, (size_t)1 << 30)); }},
      // streams created AFTER the gate was parked: the runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4)
      // hardware queues, and a kernel that lands in the parked stream's queue waits behind the gate
      {"kernel on 1 new stream", [&] { new_streams(1); }},
      {"kernels on 3 new streams", [&] { new_streams(3); }},
      {"kernels on 6 new streams", [&] { new_streams(6); }},
  };
  unsigned long long seq = 0;
  std::printf("%-36s %10s %10s\n", "foreign call", "F [s]", "L [s]");
  for (auto &p : probes) {
    CK(hipMalloc(&victim, 1 << 20));
    CK(hipHostMalloc(&hvictim, 1 << 16, hipHostMallocDefault));
    CK(hipStreamSynchronize(S));
    CK(hipStreamSynchronize(S2));
    ++seq;
    hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, S, go, seq, go + 1, 500000000ull /* 5 s at 100 MHz */);
    CK(hipGetLastError());
    const double t0 = now();
    double tF = -1, tL = -1;
    std::thread F([&] {
      const double a = now();
      p.fn();
      tF = now() - a;
    });
    std::thread L([&] {
      std::this_thread::sleep_for(std::chrono::milliseconds(300));
      const double a = now();
      hipLaunchKernelGGL(noop, dim3(1), dim3(64), 0, S, nullptr);
      tL = now() - a;
    });
    while (now() - t0 < 1.5)
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    __atomic_store_n(go, seq, __ATOMIC_RELEASE);
    F.join();
    L.join();
    CK(hipStreamSynchronize(S));
    std::printf("%-36s %10.3f %10.3f%s\n", p.name, tF, tL, go[1] ? "  (gate timed out!)" : "");
    std::fflush(stdout);
    if (victim && std::string(p.name) != "hipFree")
      CK(hipFree(victim));
    if (std::string(p.name) != "hipHostFree")
      CK(hipHostFree(hvictim));
    if (extra)
      CK(hipFree(extra));
    victim = hvictim = extra = nullptr;
  }
  return 0;
}
