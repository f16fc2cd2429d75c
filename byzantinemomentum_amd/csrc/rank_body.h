// rank_body.h — score + stable rank of the rows from the n x n squared distances, by ONE workgroup (any multiple of 64
// lanes): the body of krum_rank_kernel (pairwise.hip), also run by the last workgroup of the GATED direct kernel
// (pairwise.hip, the third launch of bm_pairwise_rank, n <= 32), so that a single-GPU Krum / Bulyan needs no rank
// launch of its own.
//
// Replaces the Python score / sort loops of aggregators/krum.py:50-62 and bulyan.py:56-69.
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

// LDS of the ranking: the distances D[n][n], the rows in ascending order S[n][n-1], the scores [n] — 2 n^2 doubles.
__host__ __device__ constexpr int rank_lds_bytes(int n) { return 2 * n * n * (int)sizeof(double); }
constexpr int kRankLdsBytes = rank_lds_bytes(BM_MAX_ROWS);  // 64 KB at n = 64

// distance as the rules see it: sqrt in fp64, non-finite -> +inf (krum.py:46-47)
__device__ __forceinline__ double rank_distance(double sq) {
  const double v = sqrt(sq);
  return (__builtin_fabs(v) < __builtin_inf()) ? v : __builtin_inf();
}

// From the distances in LDS: D[i * n + j] = rank_distance(sq[i][j]) (the diagonal is never read).  lds: rank_lds_bytes(n)
// bytes, 8-byte aligned, D first.  Every lane of the workgroup must call (barriers inside), after a barrier that made D
// visible.
//
// Every distance finds its place in its row by COUNTING — pair (i, j) counts the k with d_ik < d_ij, ties to the lower
// index — all n (n - 1) pairs at once across the workgroup (n reads of LDS each, the lanes of a wave share i and read
// the same address: broadcasts), and is scattered to S[i][place]; then lane i adds the `take` smallest of row i in
// ascending order in fp64 — the same sequence of additions as the reference's `sum(sorted(...)[:take])` — and the rows
// are ranked by score, ties to the lower index (Python's stable sort).  (Round 4 sorted each row across the 64 lanes of
// one wave with a bitonic network of 21 dependent cross-lane exchanges: 8.5 us at n = 25 and 13 us at n = 51 with 16
// waves, profiles/r05_c_full_kernel_trace.csv; counting is ~1 us at both.)
__device__ __forceinline__ void krum_rank_from_distances(double* lds, int n, int f, int m, int mode,
                                                         int32_t* __restrict__ order, double* __restrict__ scores_out) {
  const double* D = lds;
  double* S = lds + n * n;
  double* score = S + n * (n - 1);
  const int tid = threadIdx.x, threads = (int)blockDim.x;
  for (int e = tid; e < n * n; e += threads) {
    const int i = e / n, j = e - i * n;
    if (i == j) continue;
    const double v = D[e];
    const double* row = D + i * n;
    int place = 0;
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
      const double dk = row[k];
      place += (k != i && (dk < v || (dk == v && k < j))) ? 1 : 0;
    }
    S[i * (n - 1) + place] = v;
  }
  __syncthreads();
  if (tid < n) {
    // krum: n-f-1 smallest (krum.py:59-60); bulyan: m smallest (bulyan.py:58-61)
    int take = (mode == BM_RANK_KRUM) ? (n - f - 1) : m;
    if (take > n - 1) take = n - 1;
    if (take < 0) take = 0;
    double s = 0.0;
#pragma unroll 8
    for (int t = 0; t < take; ++t) s += S[tid * (n - 1) + t];  // additions stay in ascending order
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

// The same from the n x n squared distances in device memory.
__device__ __forceinline__ void krum_rank_body(const double* __restrict__ sq, int n, int f, int m, int mode,
                                               int32_t* __restrict__ order, double* __restrict__ scores_out,
                                               double* lds) {
  for (int e = threadIdx.x; e < n * n; e += (int)blockDim.x) lds[e] = rank_distance(sq[e]);
  __syncthreads();
  krum_rank_from_distances(lds, n, f, m, mode, order, scores_out);
}

}  // namespace bm
