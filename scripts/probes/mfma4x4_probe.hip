// Layout and issue-rate probe for v_mfma_f32_4x4x1_16b_f32 on gfx950 (experiment, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out /* [64 a][64 lanes][4] */) {
  const int lane = threadIdx.x;
  for (int a = 0; a < 64; ++a) {
    const float av = (lane == a) ? 1.0f : 0.0f;
    const float bv = (float)(lane + 1);
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[(a * 64 + lane) * 4 + v] = acc[v];
  }
}

__global__ void rate(float* out, long long* cycles, int iters) {
  const int lane = threadIdx.x & 63;
  float a = (float)lane, b = 1.0f;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

__global__ void rate16(float* out, long long* cycles, int iters) {
  const int lane = threadIdx.x & 63;
  float a = (float)lane, b = 1.0f;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  float* d; hipMalloc(&d, 64 * 64 * 4 * sizeof(float));
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
  float* h = (float*)malloc(64 * 64 * 4 * sizeof(float));
  hipMemcpy(h, d, 64 * 64 * 4 * sizeof(float), hipMemcpyDeviceToHost);
  // for A one-hot at lane a: which (lane, reg) are non-zero and which B lane do they carry
  for (int a = 0; a < 64; a += 1) {
    printf("A lane %2d ->", a);
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
      float x = h[(a * 64 + l) * 4 + v];
      if (x != 0.0f) printf(" D[lane %d][%d]=Blane%d", l, v, (int)x - 1);
    }
    printf("\n");
  }
  long long* dc; hipMalloc(&dc, 8 * 1024);
  float* o; hipMalloc(&o, 4096 * 4);
  for (int waves = 1; waves <= 8; waves *= 2) {
    hipLaunchKernelGGL(rate, dim3(1), dim3(64 * waves), 0, 0, o, dc, 2000);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("4x4x1_16b: %d waves/CU: %.1f clock-counter ticks per MFMA per wave\n", waves, (double)c / (2000.0 * 8));
    hipLaunchKernelGGL(rate16, dim3(1), dim3(64 * waves), 0, 0, o, dc, 2000);
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("16x16x4  : %d waves/CU: %.1f clock-counter ticks per MFMA per wave\n", waves, (double)c / (2000.0 * 8));
  }
  // wall-clock rate with the whole chip busy
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    hipEventRecord(e0);
    if (which == 0) hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, o, dc, 20000);
    else hipLaunchKernelGGL(rate16, dim3(1024), dim3(256), 0, 0, o, dc, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = 1024.0 * 4 * 20000 * 8;  // wave-level MFMAs
    double flop = which == 0 ? 512.0 : 2048.0;
    printf("%s: %.3f ms, %.1f TFLOP/s, %.2f ns per MFMA per SIMD\n", which == 0 ? "4x4x1_16b" : "16x16x4", ms,
           n * flop / ms / 1e9, ms * 1e6 / (n / 1024.0));
  }
  return 0;
}
