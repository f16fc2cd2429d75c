// store_order_probe.hip — does a result store placed BEFORE the next batch of loads delay those loads on
// gfx950 (loads and stores share the in-order vmcnt counter)?  Streams n rows (lane <-> 4 coordinates, one
// 16-byte load per row, like the column kernels), reduces them, and writes one 16-byte result per lane:
//   MODE 0: no store      MODE 1: store right after the arithmetic (before the next loads)
//   MODE 2: store of iteration i issued after the loads of iteration i+1 (first iteration peeled)
//   MODE 3: results of BATCH iterations staged in LDS, then written in one burst per wave
//   MODE 4: plain (cacheable) store     MODE 5: two column groups per iteration, results written back to back
// WORK = extra dependent FMAs per column to mimic the sorting network's VALU time.
//   hipcc --offload-arch=gfx950 -O3 -o store_order_probe store_order_probe.hip && ./store_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 25;
struct Rows { const float* p[N]; };

constexpr int BATCH = 8;
template <int MODE, int WORK>
__global__ __launch_bounds__(256, 4) void stream_kernel(Rows rows, uint32_t nvec, float* __restrict__ out) {
  __shared__ f32x4 stage[MODE == 3 ? BATCH * 256 : 1];
  const uint32_t stride = gridDim.x * 256;
  if (MODE == 3) {
    // lane l of iteration b owns column group v_b = v0 + b*stride; results staged, then written together
    uint32_t v0 = blockIdx.x * 256 + threadIdx.x;
    while (v0 < nvec) {
      int cnt = 0;
      for (int b = 0; b < BATCH; ++b) {
        const uint32_t v = v0 + b * stride;
        if (v >= nvec) break;
        f32x4 x[N];
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rows.p[i]) + v);
        f32x4 s = x[0];
#pragma unroll
        for (int i = 1; i < N; ++i) s += x[i];
#pragma unroll
        for (int w = 0; w < WORK; ++w) s = s * 1.0001f + 0.5f;
        stage[b * 256 + threadIdx.x] = s;
        ++cnt;
      }
      for (int b = 0; b < cnt; ++b)
        __builtin_nontemporal_store(stage[b * 256 + threadIdx.x], reinterpret_cast<f32x4*>(out) + v0 + b * stride);
      v0 += BATCH * stride;
    }
    return;
  }
  if (MODE == 5) {
    for (uint32_t v = (blockIdx.x * 256 + threadIdx.x) * 2; v + 1 < nvec; v += stride * 2) {
      f32x4 x[N], y[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rows.p[i]) + v);
        y[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rows.p[i]) + v + 1);
      }
      f32x4 s = x[0], t = y[0];
#pragma unroll
      for (int i = 1; i < N; ++i) { s += x[i]; t += y[i]; }
#pragma unroll
      for (int w = 0; w < WORK; ++w) { s = s * 1.0001f + 0.5f; t = t * 1.0001f + 0.5f; }
      __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(out) + v);
      __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(out) + v + 1);
    }
    return;
  }
  auto load_sum = [&](uint32_t v) {
    f32x4 x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rows.p[i]) + v);
    return x[0];  // placeholder, replaced below
  };
  (void)load_sum;
  uint32_t v = blockIdx.x * 256 + threadIdx.x;
  f32x4 pend = {0, 0, 0, 0};
  uint32_t pend_v = 0;
  bool first = true;
  for (; v < nvec; v += stride) {
    f32x4 x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rows.p[i]) + v);
    if (MODE == 2 && !first) __builtin_nontemporal_store(pend, reinterpret_cast<f32x4*>(out) + pend_v);
    f32x4 s = x[0];
#pragma unroll
    for (int i = 1; i < N; ++i) s += x[i];
#pragma unroll
    for (int w = 0; w < WORK; ++w) s = s * 1.0001f + 0.5f;
    if (MODE == 1) __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(out) + v);
    if (MODE == 4) reinterpret_cast<f32x4*>(out)[v] = s;
    if (MODE == 0 && s.x == 1.2345e-30f) __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(out) + v);
    pend = s;
    pend_v = v;
    first = false;
  }
  if (MODE == 2 && !first) __builtin_nontemporal_store(pend, reinterpret_cast<f32x4*>(out) + pend_v);
}

template <int MODE, int WORK>
static float run(const Rows& rows, uint32_t nvec, float* out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<MODE, WORK>), dim3(grid), dim3(256), 0, 0, rows, nvec, out);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<MODE, WORK>), dim3(grid), dim3(256), 0, 0, rows, nvec, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;
}

int main() {
  const int64_t d = 11173962;
  const uint32_t nvec = (uint32_t)(d / 4);
  Rows rows;
  for (int i = 0; i < N; ++i) {
    float* p;
    hipMalloc(&p, d * sizeof(float));
    hipMemset(p, 0x3c, d * sizeof(float));
    rows.p[i] = p;
  }
  float* out;
  hipMalloc(&out, d * sizeof(float));
  const double bytes = 4.0 * d * (N + 1);
  for (int grid : {16384, 4096}) {
    float t;
#define RUN(M, W) t = run<M, W>(rows, nvec, out, grid); printf("grid %5d mode %d work %3d: %7.1f us  %6.0f GB/s (algorithmic, incl. the result)\n", grid, M, W, t, bytes / t / 1e3);
    RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(3, 0) RUN(4, 0) RUN(5, 0)
    RUN(0, 200) RUN(1, 200) RUN(2, 200) RUN(3, 200) RUN(4, 200) RUN(5, 200)
  }
  return 0;
}
