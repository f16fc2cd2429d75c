// two_phase_probe.hip — can the result store of a streaming kernel be made cheap by making it GLOBALLY bursty?
// The column kernels pay 10-14 % of their time for 3.8 % of their bytes (rows_sweep_probe.hip): results trickle out
// between the reads of 256 CUs.  Here one workgroup per CU (1024 threads, all 160 KB of LDS) owns a contiguous
// 1/256 of the columns, reads `per_phase` column groups per lane while staging the results in LDS, then the whole
// workgroup writes them in one burst; all CUs started together and do equal work, so the bursts roughly coincide.
//   per_phase 10 = everything a CU produces in one burst (+ one leftover group), 5 / 2 / 1 = more, smaller bursts.
// Reference: the plain form (16 384 x 256 threads, one column group per lane, store right after the arithmetic).
//   hipcc --offload-arch=gfx950 -O3 -o two_phase_probe two_phase_probe.hip && ./two_phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 25;
struct Rows { const float* p[N]; };

__device__ __forceinline__ f32x4 column_sum(const Rows& rows, uint32_t v) {
  f32x4 x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rows.p[i]) + v);
  f32x4 s = x[0];
#pragma unroll
  for (int i = 1; i < N; ++i) s += x[i];
  return s;
}

template <bool STORE>
__global__ __launch_bounds__(256, 4) void plain_kernel(Rows rows, uint32_t nvec, float* __restrict__ out) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
    const f32x4 s = column_sum(rows, v);
    if (STORE || s.x == 1.2345e-30f) __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(out) + v);
  }
}

constexpr int kSlots = 10;  // 10 x 1024 x 16 B = 160 KB
template <bool INTERLEAVED, bool BARRIER>
__global__ __launch_bounds__(1024) void two_phase_kernel(Rows rows, uint32_t nvec, float* __restrict__ out, int per_phase) {
  __shared__ f32x4 stage[kSlots * 1024];
  const uint32_t tid = threadIdx.x;
  // contiguous: workgroup b owns [b*chunk, (b+1)*chunk); interleaved: group (it*gridDim + b)*1024 + tid
  const uint32_t chunk = (nvec + gridDim.x - 1) / gridDim.x;
  const uint32_t base = blockIdx.x * chunk;
  const uint32_t end = (base + chunk < nvec) ? base + chunk : nvec;
  const int iters = (int)((chunk + 1023) / 1024);
  auto group = [&](int it) -> uint32_t {
    if (INTERLEAVED) return ((uint32_t)it * gridDim.x + blockIdx.x) * 1024 + tid;
    return base + (uint32_t)it * 1024 + tid;
  };
  auto valid = [&](uint32_t v) { return INTERLEAVED ? (v < nvec) : (v < end); };
  for (int p0 = 0; p0 < iters; p0 += per_phase) {
    const int p1 = (p0 + per_phase < iters) ? p0 + per_phase : iters;
    for (int it = p0; it < p1; ++it) {
      const uint32_t v = group(it);
      if (valid(v)) stage[(it - p0) * 1024 + tid] = column_sum(rows, v);
    }
    if (BARRIER) __syncthreads();
    for (int it = p0; it < p1; ++it) {
      const uint32_t v = group(it);
      if (valid(v)) __builtin_nontemporal_store(stage[(it - p0) * 1024 + tid], reinterpret_cast<f32x4*>(out) + v);
    }
  }
}

template <class Launch>
static float timed(Launch&& launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;
}

int main() {
  const int64_t d = 11173962;
  const uint32_t nvec = (uint32_t)(d / 4);
  Rows rows;
  for (int i = 0; i < N; ++i) {
    float* p;
    (void)hipMalloc(&p, d * sizeof(float));
    (void)hipMemset(p, 0x3c, d * sizeof(float));
    rows.p[i] = p;
  }
  float* out;
  (void)hipMalloc(&out, d * sizeof(float));
  const double bytes = 4.0 * d * (N + 1);
  for (int round = 0; round < 2; ++round) {
    float t = timed([&] { hipLaunchKernelGGL(plain_kernel<true>, dim3(16384), dim3(256), 0, 0, rows, nvec, out); });
    printf("plain 16384x256, store        : %7.1f us  %5.0f GB/s\n", t, bytes / t / 1e3);
    t = timed([&] { hipLaunchKernelGGL(plain_kernel<false>, dim3(16384), dim3(256), 0, 0, rows, nvec, out); });
    printf("plain 16384x256, no store     : %7.1f us  %5.0f GB/s (reads only)\n", t, 4.0 * d * N / t / 1e3);
    for (int per : {10, 5, 2, 1}) {
      t = timed([&] { hipLaunchKernelGGL((two_phase_kernel<false, true>), dim3(256), dim3(1024), 0, 0, rows, nvec, out, per); });
      printf("two-phase contiguous  barrier, %2d groups per burst: %7.1f us  %5.0f GB/s\n", per, t, bytes / t / 1e3);
      t = timed([&] { hipLaunchKernelGGL((two_phase_kernel<true, true>), dim3(256), dim3(1024), 0, 0, rows, nvec, out, per); });
      printf("two-phase interleaved barrier, %2d groups per burst: %7.1f us  %5.0f GB/s\n", per, t, bytes / t / 1e3);
      t = timed([&] { hipLaunchKernelGGL((two_phase_kernel<true, false>), dim3(256), dim3(1024), 0, 0, rows, nvec, out, per); });
      printf("two-phase interleaved no barr, %2d groups per burst: %7.1f us  %5.0f GB/s\n", per, t, bytes / t / 1e3);
    }
  }
  return 0;
}
