// rank_body.h — score + stable rank of the rows from the n x n squared distances, by ONE workgroup (any multiple of 64
// lanes): the body of krum_rank_kernel (pairwise.hip), also run by the last workgroup of the GATED direct kernel
// (pairwise.hip, the third launch of bm_pairwise_rank, n <= 32), so that a single-GPU Krum / Bulyan needs no rank
// launch of its own.
//
// Replaces the Python score / sort loops of aggregators/krum.py:50-62 and bulyan.py:56-69.  The distances of a row
// are sorted across the lanes of one wave (bitonic network), then lane i adds the `take` smallest of row i in
// ascending order in fp64 — the same sequence of additions as the reference's `sum(sorted(...)[:take])` — and the
// rows are ranked by score, ties to the lower index (Python's stable sort).
#pragma once
#include "bm_common.h"

namespace bm {

// What a launch of the distance pass does on top of its own work when it belongs to bm_pairwise_rank (by value in the
// kernarg segment): rank the rows from the final distances.  `on` = 1: the last workgroup of the Gram reduction ranks
// when the accuracy gate listed nothing (the common case), the gated direct kernel when it did.
struct RankArgs {
  int on, f, m, mode;
  int32_t* order;
  double* scores;
};

constexpr int kRankSrtDoubles = BM_MAX_ROWS * (BM_MAX_ROWS + 1);  // srt[i][r] = r-th smallest distance of row i
constexpr int kRankLdsBytes = (kRankSrtDoubles + BM_MAX_ROWS) * (int)sizeof(double);

// lds: kRankLdsBytes, 8-byte aligned.  Every lane of the workgroup must call (barriers inside).
__device__ __forceinline__ void krum_rank_body(const double* __restrict__ sq, int n, int f, int m, int mode,
                                               int32_t* __restrict__ order, double* __restrict__ scores_out,
                                               double* lds) {
  double(*srt)[BM_MAX_ROWS + 1] = reinterpret_cast<double(*)[BM_MAX_ROWS + 1]>(lds);
  double* score = lds + kRankSrtDoubles;
  const double kInf = __builtin_inf();
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int waves = (int)blockDim.x >> 6;
  // One wave per row (n <= 64): lane j holds dist(i, j) = sqrt in fp64, non-finite -> +inf (krum.py:46-47), the
  // row's own lane and the lanes past n hold +inf too; the 64 values are sorted ascending across the lanes by a
  // bitonic network (21 compare-exchange steps of one cross-lane exchange each, whatever n — counting every value's
  // rank against n broadcasts, as before, was n^3 / 64 fp64 compares per stack: 2.4 x the instructions at n = 51).
  // Lanes 0 .. n-2 then hold the row's n - 1 distances in ascending order (equal values in either order: the sums
  // below do not depend on it).
  for (int i = wave; i < n; i += waves) {
    double v = kInf;
    if (lane < n && lane != i) {
      v = sqrt(sq[i * n + lane]);
      if (!(v == v) || v == kInf || v == -kInf) v = kInf;
    }
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const double o = __shfl_xor(v, j, 64);
        const bool keep_min = ((lane & k) == 0) == ((lane & j) == 0);
        v = keep_min ? __builtin_fmin(v, o) : __builtin_fmax(v, o);
      }
    }
    if (lane < n - 1) srt[i][lane] = v;
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
