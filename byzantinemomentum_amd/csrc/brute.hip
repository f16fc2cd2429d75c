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

constexpr int kBruteNodeBudgetPerWave = 1 << 18;  // search-tree nodes per wave and launch (~0.1 us each); BM_BRUTE_BUDGET overrides (tests)

struct BruteWave {
  const double* dist;  // LDS, [n][n], symmetric, non-finite where the reference's distance is
  uint64_t* stack;     // LDS, the open alternatives of the search tree (wave-uniform values)
  int n, lane;
  uint64_t adj;        // this lane's row of G(t)
  int nodes;           // search-tree nodes visited so far by this wave (all has_clique calls of the launch)
  bool exhausted;      // the budget ran out: the answers of this wave no longer mean anything
  int budget;          // search-tree nodes this wave may visit

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
      if (++nodes > budget) {
        exhausted = true;
        return false;
      }
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

// sel_out: BM_MAX_ROWS int32, the n - f selected rows ascending, then zeros.  status[0]: 0; -1 when every subset
// touches a non-finite distance (the reference then has no selection at all, brute.py:56-57,68): sel_out then holds
// n - f copies of the first row ALL of whose distances are non-finite (a gradient with a non-finite coordinate), so
// that the average that follows is non-finite where that row is instead of looking like a result; -2 when a wave
// used up its budget of search-tree nodes (kBruteNodeBudgetPerWave: the tree is exponential in f in the worst case and
// crafted distance matrices are this library's threat model — a stream must not be held for seconds): sel_out then
// holds n - f times the index -1, which the averaging kernels (reduce.hip) answer with an all-NaN vector — never a
// usable selection — and the Python host falls back to bm_brute_select, the host search, which has no such limit.
//
// ONE workgroup of kBruteWaves waves.  The three phases of the search are the host's (api.cpp), but the questions
// "does G(t) hold n - f mutually adjacent rows" are asked kBruteWaves at a time:
//   1. wave 0 asks it for the largest finite distance (is there a selection at all?), wave 1 for t = 0;
//   2. the smallest such t: every round, wave w takes the open candidate (a finite distance strictly between the
//      bounds) number (2w+1)/(2W) of the way through the row-major enumeration of the open ones — W pivots that are
//      random in VALUE — and the bounds close in on the tightest answers: the open set shrinks ~W/2-fold per round
//      (n = 25: 300 pairs, 3 rounds; n = 51: 1 275 pairs, 4 rounds) instead of 2-fold per sequential probe;
//   3. the lexicographically first set in G(t*): first the rows that lie in some such set at all (n independent
//      questions; when exactly n - f rows pass, they are the answer: the usual case), then position by position with W
//      prefixes of the lowest open rows per round, the longest prefix that extends taken whole.
// Every wave keeps the (wave-uniform) state of the search itself and recomputes the cheap parts; only the W answers
// of a round travel through LDS.  The answers are those of the sequential search: G(t) only grows with t, and the
// extraction takes the same row at every position.
constexpr int kBruteWaves = 16;

__global__ __launch_bounds__(64 * kBruteWaves) void brute_select_kernel(const double* __restrict__ sq, int n, int f,
                                                                        int32_t* __restrict__ sel_out,
                                                                        int32_t* __restrict__ status, int budget) {
  __shared__ double dist[BM_MAX_ROWS * BM_MAX_ROWS];
  __shared__ uint64_t stacks[kBruteWaves][BM_MAX_ROWS + 4];
  __shared__ double piv[kBruteWaves];
  __shared__ int res[kBruteWaves];   // 1 / 0: the answer of wave w this round; -1: it asked nothing
  __shared__ int over[kBruteWaves];  // wave w ran out of its node budget
  __shared__ uint64_t cores[kBruteWaves];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = n - f;
  // distances as the host path forms them: sqrt of the squared ones, only the [x][y], x < y entries are read
  for (int e = tid; e < n * n; e += 64 * kBruteWaves) {
    const int i = e / n, j = e - i * n;
    dist[e] = (i == j) ? 0.0 : __builtin_sqrt(sq[i < j ? i * n + j : j * n + i]);
  }
  if (lane == 0) over[wave] = 0;
  __syncthreads();
  BruteWave w{dist, stacks[wave], n, lane, 0, 0, false, budget};
  const uint64_t everyone = n == 64 ? ~(uint64_t)0 : (((uint64_t)1 << n) - 1);
  auto in_range = [&](double v, double lo, double hi) { return __builtin_fabs(v) < __builtin_inf() && v > lo && v < hi; };
  // one round of answers: wave w publishes (pivot, answer), everybody reads all of them
  auto publish = [&](double pivot, int answer) {
    if (lane == 0) {
      piv[wave] = pivot;
      res[wave] = answer;
      if (w.exhausted) over[wave] = 1;
    }
    __syncthreads();
  };
  auto any_over = [&]() {
    int o = 0;
#pragma unroll
    for (int v = 0; v < kBruteWaves; ++v) o |= over[v];
    return o != 0;
  };
  auto give_up = [&](int code, int fill) {
    if (wave == 0) {
      if (lane < BM_MAX_ROWS) sel_out[lane] = (lane < k) ? fill : 0;
      if (lane == 0) status[0] = code;
    }
  };

  // ---- 1. the largest finite distance (wave 0) and zero (wave 1) ----
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
  int answer = -1;
  if (wave < 2) {
    w.build(wave == 0 ? vmax : 0.0);
    answer = w.has_clique(everyone, k) ? 1 : 0;
  }
  publish(0.0, answer);
  const bool feasible = res[0] == 1, at_zero = res[1] == 1;
  const bool over1 = any_over();
  __syncthreads();  // (everybody has read the answers before the next round overwrites them)
  if (over1) return give_up(-2, -1);
  if (!feasible) {
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
    return give_up(-1, bad < 64 ? bad : 0);
  }

  // ---- 2. the smallest diameter.  Invariant: G(hi) holds k mutually adjacent rows, G(lo) does not ----
  double lo = 0.0, hi = at_zero ? 0.0 : vmax;
  while (!at_zero) {
    // the open candidates, counted per row (pairs i < j: lane i enumerates j > i), and their row-major prefix
    int mine = 0;
    if (lane < n)
      for (int j = lane + 1; j < n; ++j) mine += in_range(dist[lane * n + j], lo, hi) ? 1 : 0;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) break;  // nothing between lo and hi: hi is the smallest diameter
    const int askers = total < kBruteWaves ? total : kBruteWaves;
    answer = -1;
    double pivot = 0.0;
    if (wave < askers) {
      const int idx = (int)(((int64_t)(2 * wave + 1) * total) / (2 * askers));  // distinct for distinct waves
      const uint64_t owner_mask = __builtin_amdgcn_ballot_w64(incl > idx && incl - mine <= idx);
      const int src = __builtin_ctzll(owner_mask);
      if (lane == src) {
        int skip = idx - (incl - mine);
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
      answer = w.has_clique(everyone, k) ? 1 : 0;
    }
    publish(pivot, answer);
#pragma unroll
    for (int v = 0; v < kBruteWaves; ++v) {
      const int r = res[v];
      const double p = piv[v];
      if (r == 1 && p < hi) hi = p;
      if (r == 0 && p > lo) lo = p;
    }
    const bool over2 = any_over();
    __syncthreads();
    if (over2) return give_up(-2, -1);
  }

  // ---- 3. the first subset in lexicographic order in G(hi) ----
  w.build(hi);
  // 3a. the CORE: the rows that lie in some set of k mutually adjacent rows — n independent questions, wave v asks
  //     them for rows v, v + W, ...  No other row can be chosen at any position (the sequential search asks the same
  //     question with fewer rows in play), and when exactly k rows are left they ARE the answer: the usual case (the
  //     set of smallest diameter is unique), one barrier.
  uint64_t mine_core = 0;
  for (int c = wave; c < n; c += kBruteWaves)
    if (w.has_clique(readlane64(w.adj, c), k - 1)) mine_core |= (uint64_t)1 << c;
  if (lane == 0) {
    cores[wave] = mine_core;
    if (w.exhausted) over[wave] = 1;
  }
  __syncthreads();
  uint64_t cand = 0;  // rows still in play: in the core, adjacent to every chosen row, above the last chosen one
#pragma unroll
  for (int v = 0; v < kBruteWaves; ++v) cand |= cores[v];
  if (any_over()) return give_up(-2, -1);
  // 3b. position by position: a round tries the prefixes c_0, c_0 c_1, ... of the W lowest open rows — wave v takes
  //     c_0 .. c_{v-1} for chosen and asks whether c_v still extends — and the longest prefix of "yes" is taken whole;
  //     the row behind it has then failed exactly the question the sequential search would have asked at that position.
  uint64_t skipped = 0;  // rows of `cand` that were tried at this position and do not extend
  int chosen = 0;
  int32_t mine_sel = 0;
  while (chosen < k) {
    skipped &= cand;
    if (__builtin_popcountll(cand) == k - chosen && skipped == 0) {
      // what is left has exactly the size that is needed, and it holds a completion: it is the completion
      const int my = lane - chosen;
      if (my >= 0 && my < k - chosen) {
        uint64_t rest = cand;
        for (int t = 0; t < my; ++t) rest &= rest - 1;
        mine_sel = __builtin_ctzll(rest);
      }
      chosen = k;
      break;
    }
    const uint64_t open = cand & ~skipped;
    if (open == 0) break;  // (cannot happen while the invariant holds: status -1 below)
    // wave v: the state after c_0 .. c_{v-1}, then c_v
    uint64_t state = cand, rest = open;
    bool in_play = true;
    int c = 64;
    for (int t = 0; t <= wave && in_play; ++t) {
      if (rest == 0) {
        in_play = false;
        break;
      }
      c = __builtin_ctzll(rest);
      rest &= rest - 1;
      const uint64_t above = c == 63 ? 0 : ~(((uint64_t)1 << (c + 1)) - 1);
      if ((state & ((uint64_t)1 << c)) == 0) in_play = false;  // (not adjacent to an earlier row of the prefix)
      state = state & readlane64(w.adj, c) & above;
    }
    answer = -1;
    if (in_play && chosen + wave + 1 <= k) answer = w.has_clique(state, k - chosen - wave - 1) ? 1 : 0;
    publish(0.0, answer);
    int accepted = 0;  // the longest run of "yes" from wave 0 (the answers are monotone: a prefix of a prefix extends)
#pragma unroll
    for (int v = 0; v < kBruteWaves; ++v) accepted += (accepted == v && res[v] == 1) ? 1 : 0;
    const bool over3 = any_over();
    __syncthreads();
    if (over3) return give_up(-2, -1);
    // every wave replays the accepted prefix on its own copy of the state
    uint64_t walk = open;
    for (int v = 0; v < accepted; ++v) {
      const int row = __builtin_ctzll(walk);
      walk &= walk - 1;
      if (lane == chosen + v) mine_sel = row;
      const uint64_t above = row == 63 ? 0 : ~(((uint64_t)1 << (row + 1)) - 1);
      cand = cand & readlane64(w.adj, row) & above;
    }
    chosen += accepted;
    if (accepted > 0) skipped = 0;
    if (walk != 0 && accepted < kBruteWaves && chosen < k) skipped |= (uint64_t)1 << __builtin_ctzll(walk);
  }
  if (wave == 0) {
    if (lane < BM_MAX_ROWS) sel_out[lane] = (chosen == k && lane < k) ? mine_sel : 0;
    if (lane == 0) status[0] = chosen == k ? 0 : -1;
  }
}

}  // namespace bm

extern "C" int bm_brute_select_device(const double* sq_nxn, int n, int f, int32_t* sel_out, int32_t* status,
                                      void* stream) {
  using namespace bm;
  if (sq_nxn == nullptr || sel_out == nullptr || status == nullptr || n < 1 || n > BM_MAX_ROWS || f < 0 || n - f < 1)
    return BM_EINVAL;
  const int budget = tuning().brute_budget > 0 ? tuning().brute_budget : kBruteNodeBudgetPerWave;
  hipLaunchKernelGGL(brute_select_kernel, dim3(1), dim3(64 * kBruteWaves), 0, static_cast<hipStream_t>(stream), sq_nxn, n, f, sel_out,
                     status, budget);
  BM_LAUNCH_CHECK();
  return 0;
}
