// Read-only streaming micro-benchmark with the access pattern of the sample-stationary conv kernels: persistent
// workgroups, each pulling whole samples (two contiguous operands of OB and DZ bytes) with 16-byte loads and DEPTH
// samples in flight in registers.  No compute: what the memory system delivers for this pattern at a given number of
// bytes in flight.   hipcc --offload-arch=gfx950 -O3 streambench.hip -o streambench && ./streambench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                                     \
      std::exit(1);                                                                                                    \
    }                                                                                                                  \
  } while (0)

template <int DEPTH, int NT, int PA, int PB>
__global__ __launch_bounds__(NT) void stream_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, long sa,
                                                    long sb, int na, int nb, int nsamp, int chunked, uint4 *out) {
  uint4 ra[DEPTH][PA], rb[DEPTH][PB];
  const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  const int per = nsamp / nwg;
  auto sample = [&](int i) -> long { return chunked ? (long)wg * per + i : (long)wg + (long)i * nwg; };
  auto load = [&](int d, int i) {
    const long s = sample(i < per ? i : per - 1);
#pragma unroll
    for (int k = 0; k < PA; ++k) {
      int p = tid + k * NT;
      p = p < na ? p : na - 1;
      ra[d][k] = a[s * sa + p];
    }
#pragma unroll
    for (int k = 0; k < PB; ++k) {
      int p = tid + k * NT;
      p = p < nb ? p : nb - 1;
      rb[d][k] = b[s * sb + p];
    }
  };
  uint4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    load(d, d);
  for (int i = 0; i < per; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int k = 0; k < PA; ++k) {
        acc.x ^= ra[d][k].x; acc.y ^= ra[d][k].y; acc.z ^= ra[d][k].z; acc.w ^= ra[d][k].w;
      }
#pragma unroll
      for (int k = 0; k < PB; ++k) {
        acc.x ^= rb[d][k].x; acc.y ^= rb[d][k].y; acc.z ^= rb[d][k].z; acc.w ^= rb[d][k].w;
      }
      load(d, i + d + DEPTH);
    }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
    out[wg * NT + tid] = acc;
}

template <int DEPTH, int NT>
static void run(const char *name, const uint4 *a, const uint4 *b, long oba, long obb, int nsamp, int nwg, int chunked,
                uint4 *out) {
  constexpr int PA = (28224 / 16 + NT - 1) / NT, PB = (25600 / 16 + NT - 1) / NT;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 20;
  for (int w = 0; w < 3; ++w)
    stream_kernel<DEPTH, NT, PA, PB><<<nwg, NT>>>(a, b, oba / 16, obb / 16, 28224 / 16, 25600 / 16, nsamp, chunked, out);
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    stream_kernel<DEPTH, NT, PA, PB><<<nwg, NT>>>(a, b, oba / 16, obb / 16, 28224 / 16, 25600 / 16, nsamp, chunked, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)nsamp * (28224 + 25600);
  std::printf("%-28s depth %d threads %4d wgs %4d samples %6d %s: %8.1f us  %6.2f TB/s  (%5.1f KB in flight / WG)\n",
              name, DEPTH, NT, nwg, nsamp, chunked ? "chunked    " : "interleaved", us, bytes / us * 1e-6,
              DEPTH * (PA + PB) * NT * 16 / 1024.0);
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
}


// Closer to conv1 wgrad: 512-thread workgroups of which waves 0-3 load half-sample groups (14,784 + 12,800 B, 8 loads
// per thread) with DEPTH groups in flight, one barrier per group; WRITE: the loaded data also goes to LDS (x widened 2x)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int DEPTH, int WRITE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void producer_kernel(const uint4 *__restrict__ x_, const uint4 *__restrict__ dy_,
                                                       int ngroups, uint4 *out_) {
  extern __shared__ v4u lds[];
  const v4u *x = reinterpret_cast<const v4u *>(x_), *dy = reinterpret_cast<const v4u *>(dy_);
  v4u *out = reinterpret_cast<v4u *>(out_);
  const int tid = threadIdx.x, wave = tid >> 6;
  const int gs = gridDim.x;
  struct Regs {
    v4u x[4], d[4];
  };
  Regs R0, R1, R2, R3;
  v4u acc = {0, 0, 0, 0};
  auto load = [&](Regs &R, int g) {
    g = g < ngroups ? g : ngroups - 1;
    const v4u *px = x + (long)(g >> 1) * (28224 / 16) + (g & 1) * (13440 / 16);
    const v4u *pd = dy + (long)g * (12800 / 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int v = tid + 256 * k;
      R.x[k] = px[v < 924 ? v : 923];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int v = tid + 256 * k;
      R.d[k] = pd[v < 800 ? v : 799];
    }
  };
  auto use = [&](Regs &R, int buf) {
    v4u *b = lds + buf * 3072;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (WRITE != 0) {
        b[2 * (tid + 256 * k)] = R.x[k];
        b[2 * (tid + 256 * k) + 1] = R.x[k];
      } else {
        acc.x ^= R.x[k].x; acc.y ^= R.x[k].y; acc.z ^= R.x[k].z; acc.w ^= R.x[k].w;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (WRITE != 0) {
        b[2048 + tid + 256 * k] = R.d[k];
      } else {
        acc.x ^= R.d[k].x; acc.y ^= R.d[k].y; acc.z ^= R.d[k].z; acc.w ^= R.d[k].w;
      }
    }
  };
  int g = blockIdx.x;
  if (wave < 4) {
    load(R0, g);
    load(R1, g + gs);
    if constexpr (DEPTH == 4) {
      load(R2, g + 2 * gs);
      load(R3, g + 3 * gs);
    }
    for (; g < ngroups; g += DEPTH * gs) {
      use(R0, 0);
      load(R0, g + DEPTH * gs);
      __syncthreads();
      use(R1, 1);
      load(R1, g + (DEPTH + 1) * gs);
      __syncthreads();
      if constexpr (DEPTH == 4) {
        use(R2, 0);
        load(R2, g + (DEPTH + 2) * gs);
        __syncthreads();
        use(R3, 1);
        load(R3, g + (DEPTH + 3) * gs);
        __syncthreads();
      }
    }
  } else {
    for (; g < ngroups; g += DEPTH * gs) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
        __syncthreads();
    }
  }
  if (WRITE)
    acc = lds[tid];
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u)
    out[blockIdx.x * 512 + tid] = acc;
}

template <int DEPTH, int WRITE> static void runp(const uint4 *a, const uint4 *b, int nsamp, uint4 *out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 20, ngroups = 2 * nsamp;
  CK(hipFuncSetAttribute((const void *)producer_kernel<DEPTH, WRITE>, hipFuncAttributeMaxDynamicSharedMemorySize, 110000));
  for (int w = 0; w < 3; ++w)
    producer_kernel<DEPTH, WRITE><<<256, 512, 110000>>>(a, b, ngroups, out);
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    producer_kernel<DEPTH, WRITE><<<256, 512, 110000>>>(a, b, ngroups, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)nsamp * (28224 + 25600);
  std::printf("producer waves 0-3, depth %d, %s, samples %6d: %8.1f us  %6.2f TB/s\n", DEPTH,
              WRITE ? "LDS writes" : "xor only  ", nsamp, us, bytes / us * 1e-6);
}

// cold variant: every launch reads a different 4096-sample window of the 65536-sample buffers (nothing MALL-resident)
template <int DEPTH, int WRITE> static void runp_cold(const uint4 *a, const uint4 *b, uint4 *out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 16, nsamp = 4096, ngroups = 2 * nsamp;
  CK(hipFuncSetAttribute((const void *)producer_kernel<DEPTH, WRITE>, hipFuncAttributeMaxDynamicSharedMemorySize, 110000));
  producer_kernel<DEPTH, WRITE><<<256, 512, 110000>>>(a, b, ngroups, out);
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    producer_kernel<DEPTH, WRITE><<<256, 512, 110000>>>(a + (size_t)r * nsamp * (28224 / 16), b + (size_t)r * nsamp * (25600 / 16),
                                                        ngroups, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)nsamp * (28224 + 25600);
  std::printf("producer COLD windows, depth %d, %s, samples %6d: %8.1f us  %6.2f TB/s\n", DEPTH,
              WRITE ? "LDS writes" : "xor only  ", nsamp, us, bytes / us * 1e-6);
}

int main() {
  const int nmax = 65536;
  const long oba = 28224, obb = 25600;
  uint4 *a, *b, *out;
  CK(hipMalloc(&a, (size_t)nmax * oba + 4096));
  CK(hipMalloc(&b, (size_t)nmax * obb + 4096));
  CK(hipMalloc(&out, 1 << 22));
  CK(hipMemset(a, 1, (size_t)nmax * oba));
  CK(hipMemset(b, 2, (size_t)nmax * obb));
  runp_cold<2, 0>(a, b, out);
  runp_cold<4, 0>(a, b, out);
  runp_cold<4, 1>(a, b, out);
  runp_cold<4, 1>(a, b, out);
  for (int nsamp : {4096, 65536}) {
    runp<2, 0>(a, b, nsamp, out);
    runp<4, 0>(a, b, nsamp, out);
    runp<2, 1>(a, b, nsamp, out);
    runp<4, 1>(a, b, nsamp, out);
  }
  for (int nsamp : {4096}) {
    for (int chunked : {0, 1}) {
      run<1, 512>("stream", a, b, oba, obb, nsamp, 256, chunked, out);
      run<2, 512>("stream", a, b, oba, obb, nsamp, 256, chunked, out);
      run<3, 512>("stream", a, b, oba, obb, nsamp, 256, chunked, out);
      run<4, 512>("stream", a, b, oba, obb, nsamp, 256, chunked, out);
      run<6, 512>("stream", a, b, oba, obb, nsamp, 256, chunked, out);
    }
    run<1, 512>("stream 2 WG/CU", a, b, oba, obb, nsamp, 512, 0, out);
    run<2, 512>("stream 2 WG/CU", a, b, oba, obb, nsamp, 512, 0, out);
    run<2, 256>("stream 256 thr", a, b, oba, obb, nsamp, 256, 0, out);
    run<4, 256>("stream 256 thr", a, b, oba, obb, nsamp, 256, 0, out);
    run<2, 256>("stream 256 thr 4 WG/CU", a, b, oba, obb, nsamp, 1024, 0, out);
    run<1, 256>("stream 256 thr 8 WG/CU", a, b, oba, obb, nsamp, 2048, 0, out);
  }
  return 0;
}
