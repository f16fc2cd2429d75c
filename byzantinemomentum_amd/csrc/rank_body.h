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
  int on, f, m, mode, bitonic;
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
// Two ways to put a row's distances in ascending order into S (`bitonic`, wave-uniform):
//   counting (n <= 32): pair (i, j) counts the k with d_ik < d_ij, ties to the lower index — all n (n - 1) pairs at once
//     across the workgroup, n broadcast reads of LDS each — and scatters d_ij to S[i][place]: n^3 compare steps in all,
//     nothing dependent but the count itself;
//   bitonic (n > 32): one wave per row, the row's n - 1 distances (+inf in the other lanes) sorted across the 64 lanes
//     by a bitonic network (21 compare-exchange steps of one cross-lane exchange each): 21 n steps in all, a third of
//     the instructions of counting at n = 51, but 21 dependent exchanges per row.
// Then lane i adds the `take` smallest of row i in ascending order in fp64 — the same sequence of additions as the
// reference's `sum(sorted(...)[:take])` (equal values in either order: the sums do not depend on it) — and the rows are
// ranked by score, ties to the lower index (Python's stable sort).
__device__ __forceinline__ void krum_rank_from_distances(double* lds, int n, int f, int m, int mode,
                                                         int32_t* __restrict__ order, double* __restrict__ scores_out,
                                                         bool bitonic) {
  const double* D = lds;
  double* S = lds + n * n;
  double* score = S + n * (n - 1);
  const int tid = threadIdx.x, threads = (int)blockDim.x;
  if (bitonic) {
    const double kInf = __builtin_inf();
    const int wave = tid >> 6, lane = tid & 63, waves = threads >> 6;
    for (int i = wave; i < n; i += waves) {
      double v = (lane < n && lane != i) ? D[i * n + lane] : kInf;
#pragma unroll
      for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
          const double o = __shfl_xor(v, j, 64);
          const bool keep_min = ((lane & k) == 0) == ((lane & j) == 0);
          v = keep_min ? __builtin_fmin(v, o) : __builtin_fmax(v, o);
        }
      }
      if (lane < n - 1) S[i * (n - 1) + lane] = v;
    }
  } else {
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
                                               double* lds, bool bitonic) {
  for (int e = threadIdx.x; e < n * n; e += (int)blockDim.x) lds[e] = rank_distance(sq[e]);
  __syncthreads();
  krum_rank_from_distances(lds, n, f, m, mode, order, scores_out, bitonic);
}

// BM_RANK_ALGO: 0 (default) = by row count, 1 = bitonic, 2 = counting (A/B)
inline bool rank_bitonic(int n) {
  const int algo = tuning().rank_algo;
  return algo == 1 || (algo != 2 && n > 32);
}

}  // namespace bm
