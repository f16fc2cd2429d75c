// rows_sweep_probe.hip — the bare streaming rate as a function of the number of rows read per result:
// n rows (lane <-> 4 coordinates, one 16-byte non-temporal load per row), a sum, one 16-byte non-temporal
// result per lane.  The yardstick for selected_mean (m rows), bulyan pass 2 (m_max rows) and the column
// kernels (n rows): a kernel at this rate has nothing left but the memory system.
//   hipcc --offload-arch=gfx950 -O3 -o rows_sweep_probe rows_sweep_probe.hip && ./rows_sweep_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMax = 51;
struct Rows { const float* p[kMax]; };

template <int N, bool STORE, int MINW>
__global__ __launch_bounds__(256, MINW) void stream_kernel(Rows rows, uint32_t nvec, float* __restrict__ out) {
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t v = blockIdx.x * 256 + threadIdx.x; v < nvec; v += stride) {
    f32x4 x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rows.p[i]) + v);
    f32x4 s = x[0];
#pragma unroll
    for (int i = 1; i < N; ++i) s += x[i];
    if (STORE || s.x == 1.2345e-30f) __builtin_nontemporal_store(s, reinterpret_cast<f32x4*>(out) + v);
  }
}

template <int N, bool STORE, int MINW>
static float run(const Rows& rows, uint32_t nvec, float* out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<N, STORE, MINW>), dim3(grid), dim3(256), 0, 0, rows, nvec, out);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<N, STORE, MINW>), dim3(grid), dim3(256), 0, 0, rows, nvec, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;
}

template <int N, int MINW>
static void sweep(const Rows& rows, uint32_t nvec, float* out, int64_t d) {
  for (int grid : {16384, 4096, 2048}) {
    const float t1 = run<N, true, MINW>(rows, nvec, out, grid);
    const float t0 = run<N, false, MINW>(rows, nvec, out, grid);
    printf("rows %2d minw %d grid %5d: with store %7.1f us = %5.0f GB/s (incl. result) | no store %7.1f us = %5.0f GB/s | store costs %4.1f %% for %4.1f %% of the bytes\n",
           N, MINW, grid, t1, 4.0 * d * (N + 1) / t1 / 1e3, t0, 4.0 * d * N / t0 / 1e3, 100.0 * (t1 - t0) / t1, 100.0 / (N + 1));
  }
}

int main() {
  const int64_t d = 11173962;
  const uint32_t nvec = (uint32_t)(d / 4);
  Rows rows;
  for (int i = 0; i < kMax; ++i) {
    float* p;
    hipMalloc(&p, d * sizeof(float));
    hipMemset(p, 0x3c, d * sizeof(float));
    rows.p[i] = p;
  }
  float* out;
  hipMalloc(&out, d * sizeof(float));
  sweep<4, 4>(rows, nvec, out, d);
  sweep<8, 4>(rows, nvec, out, d);
  sweep<12, 4>(rows, nvec, out, d);
  sweep<18, 4>(rows, nvec, out, d);
  sweep<18, 2>(rows, nvec, out, d);
  sweep<25, 4>(rows, nvec, out, d);
  sweep<25, 2>(rows, nvec, out, d);
  sweep<37, 2>(rows, nvec, out, d);
  sweep<37, 1>(rows, nvec, out, d);
  sweep<51, 2>(rows, nvec, out, d);
  return 0;
}
