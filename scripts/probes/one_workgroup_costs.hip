// one_workgroup_costs.hip — library-free: what do the primitives of a ONE-workgroup kernel cost on an otherwise idle
// MI355X?  (attack_search_kernel, csrc/search_device.hip, took 10 us per candidate in three different forms where an
// instruction count said 1-2 us: the cost model was wrong somewhere.)
//
// Every test runs N iterations of one dependent pattern in a single workgroup and brackets it with s_memtime (shader
// clock) and wall_clock64 (constant 100 MHz): cycles per iteration and the clock the workgroup really ran at.
//   hipcc --offload-arch=gfx950 -O2 -o one_workgroup_costs one_workgroup_costs.hip && ./one_workgroup_costs
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                            \
    }                                                                                     \
  } while (0)

__device__ __forceinline__ double lane_value(double v, int src) {
  const long long bits = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

struct Stamp {
  uint64_t clk, wall;
};
__device__ __forceinline__ Stamp now() {
  Stamp s;
  s.clk = __builtin_amdgcn_s_memtime();
  s.wall = wall_clock64();
  return s;
}

// out[test] = {shader cycles, wall ticks (100 MHz), checksum bits}
__global__ __launch_bounds__(1024) void costs_kernel(int test, int iters, double seed, uint64_t* out, double* sink) {
  __shared__ double lds[4096];
  __shared__ int chase[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += blockDim.x) {
    lds[i] = seed + i;
    chase[i] = (i * 67 + 1) & 4095;
  }
  __syncthreads();
  double acc = seed * lane;
  int idx = lane;
  const Stamp t0 = now();
  switch (test) {
    case 0:  // dependent fp64 additions
      for (int i = 0; i < iters; ++i) acc += seed;
      break;
    case 1:  // fold through v_readlane: s += lane u of v
      for (int i = 0; i < iters; ++i) acc += lane_value(seed * lane, i & 63);
      break;
    case 2:  // dependent LDS reads (pointer chase)
      for (int i = 0; i < iters; ++i) idx = chase[idx];
      acc += idx;
      break;
    case 3:  // eight independent LDS reads, then eight dependent additions
      for (int i = 0; i < iters; i += 8) {
        double g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = lds[(lane * 9 + i + q) & 4095];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += g[q];
      }
      break;
    case 4:  // fp64 sqrt + division, dependent
      for (int i = 0; i < iters; ++i) acc = sqrt(acc + 2.0) / (seed + 1.5);
      break;
    case 5:  // workgroup barrier
      for (int i = 0; i < iters; ++i) __syncthreads();
      break;
    case 6:  // stable rank step: v_readlane + two compares + add
    {
      int rank = 0;
      const double v = seed * ((lane * 37) & 63);
      for (int i = 0; i < iters; ++i) {
        const double o = lane_value(v, i & 63);
        rank += (o < v || (o == v && (i & 63) < lane)) ? 1 : 0;
      }
      acc += rank;
      break;
    }
    case 7:  // a global store by one lane, then a workgroup barrier (what closes a candidate of the search kernel)
      for (int i = 0; i < iters; ++i) {
        if (tid == 0) sink[i & 255] = acc + i;
        __syncthreads();
      }
      break;
    case 8:  // ballot + popcount, dependent through the compared value
      for (int i = 0; i < iters; ++i) acc += __builtin_popcountll(__builtin_amdgcn_ballot_w64(acc + lane < 1e300));
      break;
    case 9:  // one LDS write by a lane, a barrier, a broadcast read by everybody (Y[0] of the search kernel)
      for (int i = 0; i < iters; ++i) {
        if (tid == 0) lds[0] = acc + i;
        __syncthreads();
        acc += lds[0];
        __syncthreads();
      }
      break;
    default:
      break;
  }
  const Stamp t1 = now();
  if (tid == 0) {
    out[3 * test + 0] = t1.clk - t0.clk;
    out[3 * test + 1] = t1.wall - t0.wall;
  }
  if (acc == 12345.678) sink[300] = acc;  // keep the work
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096;
  uint64_t* out;
  double* sink;
  CHECK(hipMalloc(&out, 64 * sizeof(uint64_t)));
  CHECK(hipMalloc(&sink, 512 * sizeof(double)));
  const char* names[] = {"dependent v_add_f64",
                         "s += v_readlane pair (fold across lanes)",
                         "dependent LDS read (pointer chase)",
                         "8 independent LDS reads + 8 dependent adds (per element)",
                         "sqrt + division, fp64, dependent",
                         "__syncthreads",
                         "stable-rank step (readlane pair, 2 compares, add)",
                         "global store by one lane + __syncthreads",
                         "ballot + popcount + add, dependent",
                         "LDS write, barrier, broadcast read, barrier"};
  for (int threads : {64, 1024}) {
    printf("== one workgroup of %d threads, %d iterations per test\n", threads, iters);
    for (int test = 0; test < 10; ++test) {
      uint64_t host[3] = {0, 0, 0};
      for (int rep = 0; rep < 3; ++rep) {  // (the last of three launches is reported)
        hipLaunchKernelGGL(costs_kernel, dim3(1), dim3(threads), 0, 0, test, iters, 1.25, out, sink);
        CHECK(hipDeviceSynchronize());
      }
      CHECK(hipMemcpy(host, out + 3 * test, sizeof(host), hipMemcpyDeviceToHost));
      const double cycles = (double)host[0] / iters, ns = (double)host[1] * 10.0 / iters;
      printf("  %-62s %8.1f shader cycles  %8.1f ns per iteration  (%.0f MHz)\n", names[test], cycles, ns,
             host[1] ? (double)host[0] / ((double)host[1] * 10.0) * 1e3 : 0.0);
    }
  }
  return 0;
}
