// Micro-benchmark: host <-> GPU ping-pong per "slot", two ways.
//  (a) launch per slot: host polls a ticket written by the slot's kernel, then launches the next slot's kernel;
//  (b) pre-enqueued: every slot's kernel is queued up front behind hipStreamWaitValue32 on a host-written flag; the host
//      polls the ticket and writes the next flag.
// hipcc --offload-arch=gfx950 -O3 tests/tools/waitvalue.hip -o build_tools/waitvalue
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void slot_kernel(volatile unsigned int *ticket, unsigned int t) {
  if (threadIdx.x == 0 && blockIdx.x == 0)
    __hip_atomic_store(const_cast<unsigned int *>(ticket), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (d) the slot's kernel itself waits for the release after it has published its ticket (bounded spin on the host word)
__global__ void slot_wait_kernel(volatile unsigned int *ticket, unsigned int t, const unsigned int *go, unsigned int *timeouts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __hip_atomic_store(const_cast<unsigned int *>(ticket), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < t) {
      if (wall_clock64() - t0 > 200000000ull) { // 2 s at 100 MHz
        atomicAdd(timeouts, 1u);
        break;
      }
    }
  }
}

int main() {
  const int T = 2000;
  unsigned int *ticket, *flag;
  CK(hipHostMalloc(reinterpret_cast<void **>(&ticket), 64, hipHostMallocMapped));
  CK(hipHostMalloc(reinterpret_cast<void **>(&flag), 64, hipHostMallocMapped));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  volatile unsigned int *vt = ticket;
  for (int rep = 0; rep < 2; ++rep) {
    *vt = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned int t = 1; t <= (unsigned)T; ++t) {
      hipLaunchKernelGGL(slot_kernel, dim3(1), dim3(64), 0, s, ticket, t);
      while (*vt != t) {
      }
    }
    auto t1 = std::chrono::steady_clock::now();
    std::printf("(a) launch per slot:        %.2f us per slot\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / T);
  }
  for (int rep = 0; rep < 2; ++rep) {
    *vt = 0;
    *reinterpret_cast<volatile unsigned int *>(flag) = 0;
    for (unsigned int t = 1; t <= (unsigned)T; ++t) {
      hipError_t e = hipStreamWaitValue32(s, flag, t, hipStreamWaitValueGte, 0xFFFFFFFFu);
      if (e != hipSuccess) {
        std::printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e));
        return 0;
      }
      hipLaunchKernelGGL(slot_kernel, dim3(1), dim3(64), 0, s, ticket, t);
    }
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned int t = 1; t <= (unsigned)T; ++t) {
      __atomic_store_n(flag, t, __ATOMIC_RELEASE);
      while (*vt != t) {
      }
    }
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    std::printf("(b) pre-enqueued wait-value: %.2f us per slot\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / T);
  }
  {
    unsigned int *timeouts;
    CK(hipMalloc(&timeouts, 4));
    CK(hipMemset(timeouts, 0, 4));
    __atomic_store_n(flag, 0u, __ATOMIC_RELEASE);
    for (int rep = 0; rep < 2; ++rep) {
      const unsigned base = (unsigned)rep * T; // tickets / release values base + 1 .. base + T (monotonic across reps)
      for (unsigned int t = 1; t <= (unsigned)T; ++t)
        hipLaunchKernelGGL(slot_wait_kernel, dim3(1), dim3(64), 0, s, ticket, base + t, flag, timeouts);
      auto t0 = std::chrono::steady_clock::now();
      for (unsigned int t = 1; t <= (unsigned)T; ++t) {
        while (*vt != base + t) {
        }
        __atomic_store_n(flag, base + t, __ATOMIC_RELEASE);
      }
      auto t1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s));
      unsigned int to = 0;
      CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
      std::printf("(d) the slot kernel waits for its own release: %.2f us per slot (timeouts %u)\n",
                  std::chrono::duration<double, std::micro>(t1 - t0).count() / T, to);
    }
  }
  // (c) the release word in fine-grained DEVICE memory, written by the host through the BAR (a posted PCIe write) and
  //     polled by the GPU locally - if this system maps such memory into the host's address space
  {
    unsigned int *dflag = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&dflag), 64, hipDeviceMallocFinegrained);
    hipPointerAttribute_t attr{};
    if (e != hipSuccess || hipPointerGetAttributes(&attr, dflag) != hipSuccess) {
      std::printf("(c) fine-grained device memory: not available (%s)\n", hipGetErrorString(e));
      return 0;
    }
    int host_ok = 0;
    CK(hipDeviceGetAttribute(&host_ok, hipDeviceAttributeDirectManagedMemAccessFromHost, 0));
    std::printf("(c) fine-grained device flag at %p, hostPointer %p, DirectManagedMemAccessFromHost %d\n", (void *)dflag,
                attr.hostPointer, host_ok);
    if (!attr.hostPointer) {
      std::printf("(c) no host mapping reported: skipped\n");
      return 0;
    }
    volatile unsigned int *hf = reinterpret_cast<volatile unsigned int *>(attr.hostPointer);
    CK(hipMemset(dflag, 0, 64));
    for (int rep = 0; rep < 2; ++rep) {
      *vt = 0;
      const unsigned base = rep * T;
      for (unsigned int t = 1; t <= (unsigned)T; ++t) {
        CK(hipStreamWaitValue32(s, dflag, base + t, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(slot_kernel, dim3(1), dim3(64), 0, s, ticket, t);
      }
      auto t0 = std::chrono::steady_clock::now();
      for (unsigned int t = 1; t <= (unsigned)T; ++t) {
        *hf = base + t;
        while (*vt != t) {
        }
      }
      auto t1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s));
      std::printf("(c) release word in device memory: %.2f us per slot\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / T);
    }
  }
  return 0;
}
