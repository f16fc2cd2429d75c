// gram_split.h — the pieces of the bf16 Gram contraction that more than one kernel uses: the error-free / dithered
// split of fp32 values into bf16 planes, the MFMA wrapper, the index of the compact upper triangle.
// (gram_bf16.hip: the stand-alone distance pass; step.hip: the distance pass riding along with the first pass of a step.)
#pragma once
#include "bm_common.h"

namespace bm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// row pointers travel through LDS as generic pointers; say "global" again so that the loads are
// global_load (vmcnt only) and not flat_load (vmcnt + lgkmcnt, which would tie them to the LDS waits)
typedef const float __attribute__((address_space(1)))* GlobalF;
typedef f32x4 __attribute__((address_space(1))) GlobalF4;


__host__ __device__ inline int b3_tri_index(int i, int j, int n) { return i * n - (i * (i - 1)) / 2 + (j - i); }

// two fp32 -> packed bf16 pair (round to nearest even): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// x = h + m + l exactly (h, m, l bf16): the subtractions are exact, the last residual has at most
// 8 significant bits.
__device__ __forceinline__ void split3(const f32x4 x, u32x2& h, u32x2& m, u32x2& l) {
  h.x = pack_bf16(x.x, x.y);
  h.y = pack_bf16(x.z, x.w);
  const float r0 = x.x - bf16_lo(h.x), r1 = x.y - bf16_hi(h.x);
  const float r2 = x.z - bf16_lo(h.y), r3 = x.w - bf16_hi(h.y);
  m.x = pack_bf16(r0, r1);
  m.y = pack_bf16(r2, r3);
  const float s0 = r0 - bf16_lo(m.x), s1 = r1 - bf16_hi(m.x);
  const float s2 = r2 - bf16_lo(m.y), s3 = r3 - bf16_hi(m.y);
  l.x = pack_bf16(s0, s1);
  l.y = pack_bf16(s2, s3);
}

// Two-plane form: x ~ h + m with h = rne_bf16(x) and m the remainder r = x - h rounded to bf16 STOCHASTICALLY:
// the 16 discarded bits of r are compared with 16 pseudo-random bits that depend on the COORDINATE only
// (integer add on the bit pattern, then truncation: the magnitude is rounded up with probability
// discarded/2^16, so E[m] = r exactly).  What is dropped, l = r - m, then has zero mean and is independent from
// one coordinate to the next BY CONSTRUCTION, whatever the data — with round-to-nearest the dropped part
// is a deterministic function of the value, and rows with few distinct values (constant, sign, quantised or
// sparsified gradients) turn the first-order error 2 sum_k (x_i - x_j)_k (l_i - l_j)_k of a squared distance
// into a systematic term of relative size up to 2^-16 |x| / |x_i - x_j| (3e-5 ... 5e-4 for a pair just above
// the accuracy gate) instead of a random walk sqrt(d) times smaller.  All rows share the dither of a
// coordinate, so bitwise-equal rows still give bitwise-equal planes (exact ties survive), and rows that are
// close get the same rounding direction most of the time (their l's largely cancel in l_i - l_j).
__host__ __device__ __forceinline__ unsigned dither_pair(unsigned coord) {  // (host too: tests/test_sort_network.py pins the numpy model on it)
  // two 16-bit words for coordinates coord, coord + 1 from one 32-bit mix of the (even) coordinate index
  unsigned z = coord * 0x9E3779B1u + 0x7F4A7C15u;
  z ^= z >> 15;
  z *= 0x85EBCA77u;
  z ^= z >> 13;
  z *= 0xC2B2AE3Du;
  z ^= z >> 16;
  return z;
}
__device__ __forceinline__ void split2_dithered(const f32x4 x, const unsigned d01, const unsigned d23, u32x2& h, u32x2& m) {
  h.x = pack_bf16(x.x, x.y);
  h.y = pack_bf16(x.z, x.w);
  const unsigned b0 = __builtin_bit_cast(unsigned, x.x - bf16_lo(h.x)) + (d01 & 0xffffu);
  const unsigned b1 = __builtin_bit_cast(unsigned, x.y - bf16_hi(h.x)) + (d01 >> 16);
  const unsigned b2 = __builtin_bit_cast(unsigned, x.z - bf16_lo(h.y)) + (d23 & 0xffffu);
  const unsigned b3 = __builtin_bit_cast(unsigned, x.w - bf16_hi(h.y)) + (d23 >> 16);
  // upper halves of (b1, b0) -> one packed bf16 pair: bytes {b0.2, b0.3, b1.2, b1.3}
  m.x = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  m.y = __builtin_amdgcn_perm(b3, b2, 0x07060302u);
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                 0, 0, 0);
}

}  // namespace bm
