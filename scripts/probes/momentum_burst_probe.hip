// momentum_burst_probe.hip — does the chip-wide store-burst recipe of colwise_burst_kernel help a WRITE-HEAVY pass?
// Shape of bm_momentum_stats at C5: per column group 20 sampled rows + 20 momentum buffers are read, the 20 buffers
// and 3 result vectors are written (40 reads : 23 writes, 368 B of results per lane: they cannot be staged in LDS
// for more than one iteration, so the results wait in registers for the barrier).
//   mode 0: plain — 2047 workgroups of 256 lanes, grid-stride, stores right after the arithmetic (what ships)
//   mode 1: one workgroup of 512 lanes per CU, column groups interleaved across the CUs, no barrier
//   mode 2: mode 1 + a workgroup barrier between the loads and the stores of every iteration
//   mode 3: mode 2 with 1024 lanes per CU and 8-byte columns (half the registers per lane, twice the lanes)
// WRITTEN AT THE END OF ROUND 2, NOT YET RUN (the GPU budget of the round was spent).
//   hipcc --offload-arch=gfx950 -O3 -o momentum_burst_probe momentum_burst_probe.hip && ./momentum_burst_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int H = 20;
struct Table {
  const float* g[H];
  float* b[H];
  float* out[3];
};

template <class V>
__device__ __forceinline__ void one_group(const Table& t, uint32_t v, V (&nb)[H], V (&res)[3]) {
  V g[H], b[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    g[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(t.g[i]) + v);
    b[i] = __builtin_nontemporal_load(reinterpret_cast<const V*>(t.b[i]) + v);
  }
  V sa = g[0], sb;
#pragma unroll
  for (int i = 0; i < H; ++i) nb[i] = b[i] * 0.99f + g[i] * 0.01f;
  sb = nb[0];
#pragma unroll
  for (int i = 1; i < H; ++i) {
    sa += g[i];
    sb += nb[i];
  }
  res[0] = sa * (1.0f / H);
  res[1] = sb * (1.0f / H);
  res[2] = res[1] * -0.1f;
}

template <class V>
__device__ __forceinline__ void store_group(const Table& t, uint32_t v, const V (&nb)[H], const V (&res)[3]) {
#pragma unroll
  for (int i = 0; i < H; ++i) __builtin_nontemporal_store(nb[i], reinterpret_cast<V*>(t.b[i]) + v);
#pragma unroll
  for (int i = 0; i < 3; ++i) __builtin_nontemporal_store(res[i], reinterpret_cast<V*>(t.out[i]) + v);
}

template <class V, int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void probe_kernel(Table t, uint32_t nvec) {
  const uint32_t span = gridDim.x * THREADS;
  if (MODE == 0) {
    for (uint32_t v = blockIdx.x * THREADS + threadIdx.x; v < nvec; v += span) {
      V nb[H], res[3];
      one_group<V>(t, v, nb, res);
      store_group<V>(t, v, nb, res);
    }
    return;
  }
  const uint32_t iters = (nvec + span - 1) / span;
  for (uint32_t it = 0; it < iters; ++it) {
    const uint32_t v = it * span + blockIdx.x * THREADS + threadIdx.x;
    V nb[H], res[3];
    if (v < nvec) one_group<V>(t, v, nb, res);
    if (MODE >= 2) __syncthreads();
    if (v < nvec) store_group<V>(t, v, nb, res);
  }
}

template <class Launch>
static float timed(Launch&& launch) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) launch();
  (void)hipEventRecord(a);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;
}

int main() {
  const int64_t d = 36546980;
  Table t;
  for (int i = 0; i < H; ++i) {
    float *g, *b;
    (void)hipMalloc(&g, d * sizeof(float));
    (void)hipMalloc(&b, d * sizeof(float));
    (void)hipMemset(g, 0x3c, d * sizeof(float));
    (void)hipMemset(b, 0, d * sizeof(float));
    t.g[i] = g;
    t.b[i] = b;
  }
  for (int i = 0; i < 3; ++i) (void)hipMalloc(&t.out[i], d * sizeof(float));
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const double bytes = 4.0 * d * (3 * H + 3);
  const uint32_t nv4 = (uint32_t)(d / 4), nv2 = (uint32_t)(d / 2);
  for (int round = 0; round < 2; ++round) {
    float us = timed([&] { hipLaunchKernelGGL((probe_kernel<f32x4, 256, 0>), dim3(2047), dim3(256), 0, 0, t, nv4); });
    printf("mode 0 plain 2047x256, 16-byte columns           : %8.1f us  %5.0f GB/s\n", us, bytes / us / 1e3);
    us = timed([&] { hipLaunchKernelGGL((probe_kernel<f32x4, 512, 1>), dim3(cus), dim3(512), 0, 0, t, nv4); });
    printf("mode 1 one 512-lane workgroup per CU, interleaved : %8.1f us  %5.0f GB/s\n", us, bytes / us / 1e3);
    us = timed([&] { hipLaunchKernelGGL((probe_kernel<f32x4, 512, 2>), dim3(cus), dim3(512), 0, 0, t, nv4); });
    printf("mode 2 = mode 1 + barrier before the stores       : %8.1f us  %5.0f GB/s\n", us, bytes / us / 1e3);
    us = timed([&] { hipLaunchKernelGGL((probe_kernel<f32x2, 1024, 2>), dim3(cus), dim3(1024), 0, 0, t, nv2); });
    printf("mode 3 = mode 2, 1024 lanes, 8-byte columns       : %8.1f us  %5.0f GB/s\n", us, bytes / us / 1e3);
  }
  return 0;
}
