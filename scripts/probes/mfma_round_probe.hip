// mfma_round_probe.hip — how does v_mfma_f32_16x16x32_bf16 round D = C + sum_k a_k b_k when the products are far
// below ulp(C)?  (gram_bf16.hip adds the m*m products, 2^-18 of h*h, into the accumulator of the h*h products.)
// One wave; every A row / B column holds the same bf16 value, so every D element is C + 32 * a * b.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_round_probe mfma_round_probe.hip && ./mfma_round_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned a_bits, unsigned b_bits, float c, float* out) {
  const unsigned pa = a_bits | (a_bits << 16), pb = b_bits | (b_bits << 16);
  const u32x4 A = {pa, pa, pa, pa}, B = {pb, pb, pb, pb};
  const f32x4 C = {c, c, c, c};
  const f32x4 D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = D[0];
}

static float bf16_to_float(unsigned b) {
  unsigned u = b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4);
  int up = 0, down = 0, rne_match = 0, total = 0;
  for (int t = 0; t < 64; ++t) {
    const unsigned a_bits = 0x3f80 + (t * 7) % 128;          // 1.0 .. 2.0, 8-bit mantissas
    const unsigned b_bits = 0x3a00 + (t * 13) % 128;         // ~ 2^-11
    const float c = 1.0f + 0.37f * t;                         // accumulator far above the products
    for (int sign = 0; sign < 2; ++sign) {
      const unsigned bb = b_bits | (sign ? 0x8000u : 0u);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, a_bits, bb, c, out);
      float got;
      (void)hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
      const double exact = (double)c + 32.0 * (double)bf16_to_float(a_bits) * (double)bf16_to_float(bb);
      const float rne = (float)exact;
      const float lo = nextafterf(rne, -INFINITY), hi = nextafterf(rne, INFINITY);
      ++total;
      if (got == rne) ++rne_match;
      if ((double)got > exact) ++up;
      if ((double)got < exact) ++down;
      if (t < 6) printf("c=%.7g p32=%.7g exact=%.12g got=%.9g rne=%.9g (lo %.9g hi %.9g)\n", c, exact - c, exact, got, rne, lo, hi);
    }
  }
  printf("total %d: equals RNE %d, above exact %d, below exact %d\n", total, rne_match, up, down);
  return 0;
}
