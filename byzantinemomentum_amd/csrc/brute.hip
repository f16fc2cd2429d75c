// brute.hip — the subset search of the Brute rule ON THE DEVICE (aggregators/brute.py:47-68).
//
// bm_brute_select (api.cpp) answers from the host: the caller copies the n x n matrix out (one synchronisation),
// searches, and copies the n - f indices back in — two round trips in the middle of a rule that is otherwise two
// d-sized kernels, and nothing a HIP graph can record.  Here ONE wave does the same search on the squared distances
// where they are, and writes the index table the averaging kernel already reads from device memory
// (bm_selected_mean): distances -> search -> mean is three launches on the caller's stream, no host involvement.
//
// Same algorithm, same answer as the host search (which the CPU tests pin on exhaustive enumeration):
//   * G(t) = {pairs at finite distance <= t}; "n - f mutually adjacent rows among `cand`" is decided by the same
//     search tree (take the row with the most non-neighbours, the lowest index among equals: either it stays and
//     they all go, or it goes), depth <= f, as an explicit stack of 64-bit row sets, plus one reduction rule (rows
//     with more non-neighbours than removals left go at once) — the truth of the question does not depend on how the
//     tree is walked;
//   * the smallest t among {0} and the finite distances for which G(t) holds such a set: the host bisects over the
//     SORTED distances; a wave has no cheap sort of up to 2 016 doubles, so it bisects quickselect-fashion — the
//     pivot is the middle open candidate of the middle row that still has one — which visits O(log) pivots on any
//     input that is not built against it and stops at the same t (G only grows with t);
//   * the lexicographically first such set in G(t*): position by position, the smallest row that still extends.
// Lane i owns row i: its adjacency set, its count of non-neighbours in `cand`; the row sets themselves are wave-uniform.
#include "bm_common.h"

namespace bm {

namespace {

// lane `src` (wave-uniform) of a 64-bit per-lane value, through v_readlane (no LDS crossbar, no latency chain)
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ double readlane_f64(double v, int src) {
  return __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(v), src));
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

struct BruteWave {
  const double* dist;  // LDS, [n][n], symmetric, non-finite where the reference's distance is
  uint64_t* stack;     // LDS, the open alternatives of the search tree (wave-uniform values)
  int n, lane;
  uint64_t adj;        // this lane's row of G(t)

  __device__ void build(double t) {
    uint64_t a = 0;
    if (lane < n) {
      for (int j = 0; j < n; ++j) {
        const double v = dist[lane * n + j];
        if (j != lane && __builtin_fabs(v) < __builtin_inf() && v <= t) a |= (uint64_t)1 << j;
      }
    }
    adj = a;
  }

  // are there `need` mutually adjacent rows among `cand`?  (wave-uniform arguments and result)
  // Everything that steers the search is wave-uniform and lives on the scalar unit: the row sets, the stack pointer,
  // the worst row (six ballots over the bits of the per-lane counts, ties to the lowest lane: the host loop keeps the
  // first maximum), its adjacency row (v_readlane).  The alternative tried first stays in registers; only the one
  // tried second goes through the LDS stack.
  __device__ bool has_clique(uint64_t cand, int need) {
    int top = 0;
    bool have = true;
    for (;;) {
      if (!have) {
        if (top == 0) return false;
        cand = stack[--top];
      }
      have = false;
      const int count = __builtin_popcountll(cand);
      if (count < need) continue;
      if (need <= 1) return true;
      const uint64_t me = (uint64_t)1 << lane;
      const int missing = (cand & me) ? __builtin_popcountll(cand & ~adj & ~me) : 0;
      uint64_t active = __builtin_amdgcn_ballot_w64(missing > 0);
      if (active == 0) return true;  // no non-adjacent pair left: `cand` itself, count >= need rows
      const int budget = count - need;
      if (budget == 0) continue;
      // rows that cannot stay (keeping one would cost more removals than the budget allows) all go at once, without
      // branching: the answer does not depend on the order in which the tree is explored, only its size does — on the
      // stacks of SURVEY 8d this rule alone settles most probes (50 instead of 120 tree nodes at n = 25, f = 5)
      const uint64_t forced = __builtin_amdgcn_ballot_w64(missing > budget);
      if (forced != 0) {
        cand &= ~forced;
        have = true;
        continue;
      }
#pragma unroll
      for (int b = 5; b >= 0; --b) {
        const uint64_t with_bit = __builtin_amdgcn_ballot_w64(((missing >> b) & 1) != 0) & active;
        active = with_bit != 0 ? with_bit : active;
      }
      const int worst = __builtin_ctzll(active);
      const int worst_missing = __builtin_amdgcn_readlane(missing, worst);
      const uint64_t bit = (uint64_t)1 << worst;
      const uint64_t goes = cand & ~bit;
      if (worst_missing <= budget) {
        stack[top++] = goes;                               // it goes (tried second)
        cand = cand & (readlane64(adj, worst) | bit);      // it stays, they go (tried first)
      } else {
        cand = goes;
      }
      have = true;
    }
  }
};

}  // namespace

// sel_out: BM_MAX_ROWS int32, the n - f selected rows ascending, then zeros.  status[0]: 0, or -1 when every subset
// touches a non-finite distance (the reference then has no selection at all, brute.py:56-57,68): sel_out then holds
// n - f copies of the first row ALL of whose distances are non-finite (a gradient with a non-finite coordinate), so
// that the average that follows is non-finite where that row is instead of looking like a result.
__global__ __launch_bounds__(64) void brute_select_kernel(const double* __restrict__ sq, int n, int f,
                                                          int32_t* __restrict__ sel_out, int32_t* __restrict__ status) {
  __shared__ double dist[BM_MAX_ROWS * BM_MAX_ROWS];
  __shared__ uint64_t stack[BM_MAX_ROWS + 4];
  const int lane = threadIdx.x;
  const int k = n - f;
  // distances as the host path forms them: sqrt of the squared ones, only the [x][y], x < y entries are read
  for (int e = lane; e < n * n; e += 64) {
    const int i = e / n, j = e - i * n;
    dist[e] = (i == j) ? 0.0 : __builtin_sqrt(sq[i < j ? i * n + j : j * n + i]);
  }
  __syncthreads();
  BruteWave w{dist, stack, n, lane, 0};
  const uint64_t everyone = n == 64 ? ~(uint64_t)0 : (((uint64_t)1 << n) - 1);

  // candidates: 0 and every finite distance > 0 (pairs i < j: lane i enumerates j > i)
  auto in_range = [&](double v, double lo, double hi) { return __builtin_fabs(v) < __builtin_inf() && v > lo && v < hi; };
  double vmax = 0.0;
  if (lane < n)
    for (int j = lane + 1; j < n; ++j) {
      const double v = dist[lane * n + j];
      if (__builtin_fabs(v) < __builtin_inf() && v > vmax) vmax = v;
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(vmax, off, 64);
    vmax = o > vmax ? o : vmax;
  }
  w.build(vmax);
  if (!w.has_clique(everyone, k)) {
    // a row whose distances are ALL non-finite (a gradient with a non-finite coordinate); else one that has any
    int all_bad = 64, any_bad = 64;
    if (lane < n && n > 1) {
      int count = 0;
      for (int j = 0; j < n; ++j)
        if (j != lane && !(__builtin_fabs(dist[lane * n + j]) < __builtin_inf())) ++count;
      if (count == n - 1) all_bad = lane;
      if (count > 0) any_bad = lane;
    }
    all_bad = -wave_max_i32(-all_bad);
    any_bad = -wave_max_i32(-any_bad);
    const int bad = all_bad < 64 ? all_bad : any_bad;
    if (lane < BM_MAX_ROWS) sel_out[lane] = (lane < k && bad < 64) ? bad : 0;
    if (lane == 0) status[0] = -1;
    return;
  }
  // invariant: G(hi) holds k mutually adjacent rows, G(lo) does not (lo = -1: nothing is known below 0)
  double lo = -1.0, hi = vmax;
  w.build(0.0);
  if (w.has_clique(everyone, k)) {
    hi = 0.0;
  } else {
    lo = 0.0;
    for (;;) {
      // the open candidates, counted per row; the pivot: the middle one of the middle row that has any (ballots and
      // v_readlane only: no cross-lane scan)
      int mine = 0;
      if (lane < n)
        for (int j = lane + 1; j < n; ++j) mine += in_range(dist[lane * n + j], lo, hi) ? 1 : 0;
      const uint64_t holders = __builtin_amdgcn_ballot_w64(mine > 0);
      if (holders == 0) break;  // nothing between lo and hi: hi is the smallest diameter
      const int my_rank = __builtin_popcountll(holders & (((uint64_t)1 << lane) - 1));
      const uint64_t owner_mask = __builtin_amdgcn_ballot_w64(mine > 0 && my_rank == __builtin_popcountll(holders) / 2);
      const int src = __builtin_ctzll(owner_mask);
      double pivot = 0.0;
      if (lane == src) {
        int skip = mine / 2;
        for (int j = lane + 1; j < n; ++j) {
          const double v = dist[lane * n + j];
          if (in_range(v, lo, hi)) {
            if (skip == 0) {
              pivot = v;
              break;
            }
            --skip;
          }
        }
      }
      pivot = readlane_f64(pivot, src);
      w.build(pivot);
      if (w.has_clique(everyone, k))
        hi = pivot;
      else
        lo = pivot;
    }
  }
  w.build(hi);
  // the first subset in lexicographic order: the smallest row that still leaves a completion among the rows above it
  uint64_t cand = everyone;
  int chosen = 0;
  int32_t mine_sel = 0;
  for (int c = 0; c < n && chosen < k; ++c) {
    const uint64_t bit = (uint64_t)1 << c;
    if ((cand & bit) == 0) continue;
    const uint64_t above = c == 63 ? 0 : ~(((uint64_t)1 << (c + 1)) - 1);
    const uint64_t next = cand & readlane64(w.adj, c) & above;
    if (w.has_clique(next, k - chosen - 1)) {
      if (lane == chosen) mine_sel = c;
      ++chosen;
      cand = next;
    }
  }
  if (lane < BM_MAX_ROWS) sel_out[lane] = lane < chosen ? mine_sel : 0;
  if (lane == 0) status[0] = chosen == k ? 0 : -1;
}

}  // namespace bm

extern "C" int bm_brute_select_device(const double* sq_nxn, int n, int f, int32_t* sel_out, int32_t* status,
                                      void* stream) {
  using namespace bm;
  if (sq_nxn == nullptr || sel_out == nullptr || status == nullptr || n < 1 || n > BM_MAX_ROWS || f < 0 || n - f < 1)
    return BM_EINVAL;
  hipLaunchKernelGGL(brute_select_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sq_nxn, n, f, sel_out,
                     status);
  BM_LAUNCH_CHECK();
  return 0;
}
