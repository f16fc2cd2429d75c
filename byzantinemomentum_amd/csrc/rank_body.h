// rank_body.h — score + stable rank of the rows from the n x n squared distances, by ONE workgroup (any multiple of 64
// lanes): the body of krum_rank_kernel (pairwise.hip), also run by the last workgroup of the Gram kernel when the
// accuracy gate lists nothing (gram_bf16.hip), so that a single-GPU Krum / Bulyan needs no rank launch of its own.
//
// Replaces the Python score / sort loops of aggregators/krum.py:50-62 and bulyan.py:56-69.  Distances of a row are
// ranked by counting, then lane i adds the `take` smallest of row i in ascending order in fp64 — the same sequence of
// additions as the reference's `sum(sorted(...)[:take])` — and the rows are ranked by score, ties to the lower index
// (Python's stable sort).
#pragma once
#include "bm_common.h"

namespace bm {

constexpr int kRankSrtDoubles = BM_MAX_ROWS * (BM_MAX_ROWS + 1);  // srt[i][r] = r-th smallest distance of row i
constexpr int kRankLdsBytes = (kRankSrtDoubles + BM_MAX_ROWS) * (int)sizeof(double);

__device__ __forceinline__ double rank_readlane_f64(double v, int lane) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)bits, lane);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// lds: kRankLdsBytes, 8-byte aligned.  Every lane of the workgroup must call (barriers inside).
__device__ __forceinline__ void krum_rank_body(const double* __restrict__ sq, int n, int f, int m, int mode,
                                               int32_t* __restrict__ order, double* __restrict__ scores_out,
                                               double* lds) {
  double(*srt)[BM_MAX_ROWS + 1] = reinterpret_cast<double(*)[BM_MAX_ROWS + 1]>(lds);
  double* score = lds + kRankSrtDoubles;
  const double kInf = __builtin_inf();
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int waves = (int)blockDim.x >> 6;
  // One wave per row (n <= 64): lane j holds dist(i, j) = sqrt in fp64, non-finite -> +inf (krum.py:46-47); its rank
  // among the row's other distances is counted against every lane's value broadcast from its register (v_readlane).
  for (int i = wave; i < n; i += waves) {
    const bool mine = lane < n && lane != i;
    double v = kInf;
    if (mine) {
      v = sqrt(sq[i * n + lane]);
      if (!(v == v) || v == kInf || v == -kInf) v = kInf;
    }
    int rank = 0;
    for (int l = 0; l < n; ++l) {  // wave-uniform
      const double o = rank_readlane_f64(v, l);
      rank += (l != i && (o < v || (o == v && l < lane))) ? 1 : 0;
    }
    if (mine) srt[i][rank] = v;
  }
  __syncthreads();
  if (tid < n) {
    // krum: n-f-1 smallest (krum.py:59-60); bulyan: m smallest (bulyan.py:58-61)
    int take = (mode == BM_RANK_KRUM) ? (n - f - 1) : m;
    if (take > n - 1) take = n - 1;
    if (take < 0) take = 0;
    double s = 0.0;
#pragma unroll 8
    for (int t = 0; t < take; ++t) s += srt[tid][t];  // additions stay in ascending order
    score[tid] = s;
    if (scores_out != nullptr) scores_out[tid] = s;
  }
  __syncthreads();
  if (tid < n) {
    // stable argsort: rank = #rows with a smaller score, ties to the lower index
    const double si = score[tid];
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < n; ++j) {
      const double sj = score[j];
      rank += (sj < si || (sj == si && j < tid)) ? 1 : 0;
    }
    order[rank] = tid;
  }
}

}  // namespace bm
