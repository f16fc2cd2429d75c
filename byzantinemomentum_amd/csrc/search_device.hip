// search_device.hip — the attacks' factor search (attacks/identical.py:67-77 with tools/misc.py:468-514, the
// reference's default factor=-16) evaluated ON THE DEVICE for the rules whose output is the mean of a selected subset
// (Multi-Krum, Average): bm_attack_line_search_device.
//
// The host form (linesearch.cpp, bm_attack_line_search) copies the (h+2) x (h+2) squared distances of ONE distance
// pass to the host and evaluates the candidates there: a D2H copy, a stream synchronisation and ~0.17 ms of host
// arithmetic in the middle of every step (per_gar.attack_search_c3_krum.legs).  Here one workgroup does the same
// arithmetic where the distances already are; the factor stays in device memory and the Byzantine vector is formed from
// it (bm_multi_fma3_bdev): no copy, no synchronisation, capturable in a graph.  Same candidates, same bits: the cursor
// and the closed forms are the host form's (search_core.h); tests/test_gpu_search_device.py compares factor and trace
// bit for bit.
//
// Per candidate t the n x n distances of honests + [avg + t att] * k differ from those among the honest rows only in
// the Byzantine row / column, so the ranking of krum.py:44-62 is not recomputed from scratch:
//   * once: <u_i, u_j>, and every honest row's distances to the other honest rows in ascending order (a bitonic
//     network per row, sixteen waves);
//   * per candidate: lane i of wave 0 forms dq_i = |h_i - byz(t)|, finds by binary search how many of its row's sorted
//     distances lie below it and adds the `take` smallest of the merged sequence (those, k copies of dq_i, the rest) in
//     ascending order (the additions of rank_order() on the host: equal values in either order give the same sums);
//     wave 1 does the Byzantine row meanwhile (all k are the same row: k - 1 zeros, then the dq in ascending order, their
//     stable rank counted by all sixteen waves); the stable argsort of the n scores is counted by all sixteen waves, four
//     rows each; wave 0 turns the selected set (a ballot) into the objective (row sums in index order, then their sum
//     over the lanes in the butterfly order of search_core.h).
//   Cost and history: profiles/r06_device_search.txt (the reference's semantics make every score a SEQUENTIAL sum and a
//   loop iteration around one dependent fp64 addition is 32 cycles on this chip, 64 when a v_readlane pair feeds it:
//   scripts/probes/one_workgroup_costs.hip).
#include <cstdlib>

#include "bm_common.h"
#include "rank_body.h"
#include "search_core.h"

namespace bm {

// One workgroup of 16 waves.  All of them share the set-up (a wave sorts up to five honest rows) and the two stable
// ranks of a candidate (four rows' worth of comparisons each); wave 0 does the honest rows' scores (a row per lane) and
// the objective, wave 1 the Byzantine row meanwhile.  Three or four workgroup barriers per candidate.
constexpr int kSearchBlock = 1024;
constexpr int kSearchWaves = kSearchBlock / 64;
constexpr int kSortRowsPerWave = BM_MAX_ROWS / kSearchWaves;  // 4: row i is sorted by wave i % 16
constexpr int kRankChunk = BM_MAX_ROWS / kSearchWaves;        // rows whose values one wave compares with everybody's
constexpr int kByzWave = 1;

// Row length of UU and HS: the rows are read in groups of eight (span = h rounded up), and odd, so that lane i walking row i
// meets no bank conflict.
__host__ __device__ inline int search_ld(int h) { return attack_row_span(h) + 1; }
// LDS: UU[h][ld] (<u_i, u_j>), HS[h][ld] (row i's distances to the other honest rows, ascending), SC[64] (scores), Q[64]
// (sorted dq), Y[2] (objective), PART / PARTQ[16][64] (partial ranks of the scores / of the dq, int)
__host__ __device__ inline size_t search_lds_bytes(int h) {
  return (size_t)(2 * h * search_ld(h) + 2 * BM_MAX_ROWS + 2) * sizeof(double) +
         (size_t)2 * kSearchWaves * BM_MAX_ROWS * sizeof(int);
}

__device__ __forceinline__ double lane_value(double v, int src) {  // lane `src` (wave-uniform) of v, through v_readlane
  const long long bits = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// How many of the lanes first .. first + kRankChunk - 1 (below `count`) hold a smaller value than this lane's, ties to the
// lower lane: one wave's share of a stable ascending rank over `count` lanes.
__device__ __forceinline__ int rank_share(double v, int lane, int first, int count) {
  int before = 0;
#pragma unroll
  for (int q = 0; q < kRankChunk; ++q) {
    const int j = first + q;  // (wave-uniform)
    const double o = lane_value(v, j < count ? j : 0);
    before += (j < count && (o < v || (o == v && j < lane))) ? 1 : 0;
  }
  return before;
}

// The largest v (0 <= v <= 63) over the lanes of the wave, as a wave-uniform number: `limit` at once when some lane has
// reached it (the common case of the stretches below), else bit by bit with six ballots.
__device__ __forceinline__ int longest(int v, int limit) {
  if (__builtin_amdgcn_ballot_w64(v >= limit) != 0ull) return limit;
  unsigned long long among = ~0ull;
  int most = 0;
#pragma unroll
  for (int bit = 5; bit >= 0; --bit) {
    const unsigned long long with = __builtin_amdgcn_ballot_w64(((v >> bit) & 1) != 0) & among;
    if (with != 0ull) {
      among = with;
      most |= 1 << bit;
    }
  }
  return most;
}

// The sum of one value per lane over all 64 lanes in a FIXED order every lane can follow at once: v <- v + (v of lane
// ^ 1), then ^ 2, ^ 4, ... ^ 32 (fp64 addition commutes, so both partners of an exchange form the same sum and all lanes
// end with the same bits).  The host form adds in the same order (search_core.h, butterfly_order_sum).
__device__ __forceinline__ double butterfly_sum(double v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) v = v + __shfl_xor(v, s, 64);
  return v;
}

// TRACE (measurement only, BM_SEARCH_TRACE=1 in the environment at the call): waves 0 and 1 also write the shader clock
// at the phase boundaries of every candidate behind the results — out[1 + 2 evals + (2 e + wave) * kTraceSlots + slot],
// cycles since the kernel started (scripts/search_kernel_probe.py prints them).
constexpr int kTraceSlots = 12;
template <bool TRACE, bool RANKING = false>
__global__ __launch_bounds__(kSearchBlock) void attack_search_kernel(const double* __restrict__ ext, int h, int k, int f,
                                                                     int rule, int m, int evals, int negative,
                                                                     double* __restrict__ out, const double* __restrict__ t_dev,
                                                                     int32_t* __restrict__ order_out, int take_arg) {
  // RANKING MODE (the RANKING instance; bm_attack_ranking_device): ONE candidate, its factor read from device memory, and
  // instead of the objective the stable ranking of the n scores — order_out[r] = the row of rank r, padded with zeros to
  // 64 entries: what bm_krum_rank would give for honests + [avg + t att] * k (Bulyan's searches rank with it; take_arg =
  // the number of distances a score adds, m for Bulyan's ranking, bulyan.py:48-62).
  constexpr bool ranking = RANKING;  // (its own instance: the search instances carry none of it)
  const unsigned long long clock0 = TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
  extern __shared__ double search_smem[];
  const int n = h + k, e = h + 2, tid = threadIdx.x, ld = search_ld(h);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  auto stamp = [&](int ev, int slot) {
    if (TRACE && lane == 0 && wave <= kByzWave)
      out[1 + 2 * evals + (2 * ev + wave) * kTraceSlots + slot] = (double)(__builtin_amdgcn_s_memtime() - clock0);
  };
  double* const UU = search_smem;
  double* const HS = UU + h * ld;
  double* const SC = HS + h * ld;
  double* const Q = SC + BM_MAX_ROWS;
  double* const Y = Q + BM_MAX_ROWS;
  int* const PART = reinterpret_cast<int*>(Y + 2);
  int* const PARTQ = PART + kSearchWaves * BM_MAX_ROWS;
  const double c = ext[h * e + h + 1];
  const double kInf = __builtin_inf();
  const bool krum = rule == BM_RULE_KRUM;

  // ---- once: <u_i, u_j> (everybody), and every honest row's distances to the other honest rows in ascending order (a
  // 64-lane bitonic network per row, the four rows of a wave going through it together so that their exchanges overlap)
  for (int p = tid; p < h * h; p += kSearchBlock) {
    const int i = p / h, j = p - i * h;
    UU[i * ld + j] = attack_uu(ext[i * e + h], ext[j * e + h], ext[i * e + j], i == j);
  }
  if (krum) {
    double sorted[kSortRowsPerWave];
#pragma unroll
    for (int r = 0; r < kSortRowsPerWave; ++r) {
      const int i = wave + r * kSearchWaves;
      sorted[r] = (i < h && lane < h && lane != i) ? rank_distance(ext[i * e + lane]) : kInf;
    }
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
      for (int j = kk >> 1; j > 0; j >>= 1) {
        const bool keep_min = ((lane & kk) == 0) == ((lane & j) == 0);
#pragma unroll
        for (int r = 0; r < kSortRowsPerWave; ++r) {
          const double o = __shfl_xor(sorted[r], j, 64);
          sorted[r] = keep_min ? __builtin_fmin(sorted[r], o) : __builtin_fmax(sorted[r], o);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kSortRowsPerWave; ++r) {
      const int i = wave + r * kSearchWaves;
      if (i < h && lane < h - 1) HS[i * ld + lane] = sorted[r];
    }
  }
  // lane i < h is honest row i (its dq in every wave, its score and its part of the objective in wave 0)
  const bool honest = lane < h;
  const double a = honest ? ext[lane * e + h] : 0.0;
  const double w = honest ? attack_w(a, c, ext[lane * e + h + 1]) : 0.0;
  const double* const uu = UU + (honest ? lane : 0) * ld;
  const double* const hs = HS + (honest ? lane : 0) * ld;
  int take = take_arg >= 0 ? take_arg : n - f - 1;  // krum.py:59-60
  take = take > n - 1 ? n - 1 : take;
  take = take < 0 ? 0 : take;
  const int count = krum ? m : n;
  const int hm1 = h - 1, last = hm1 > 0 ? hm1 - 1 : 0, span = attack_row_span(h);
  const unsigned long long honest_rows = (h >= 64) ? ~0ull : ((1ull << h) - 1ull);
  unsigned long long selected = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);  // Average: every row, at every candidate
  double row = 0.0;
  __syncthreads();
  stamp(0, 11);

  bm_search cur;
  cursor_begin(&cur, 0.0, 1.0, 0.8);  // the attack's call: tools.line_maximize(eval_factor, evals=evals)
  for (int ev = 0; ev < evals; ++ev) {  // (every lane runs the cursor: the same values everywhere)
    cursor_propose(&cur);
    const double x = cur.probe;
    const double t = ranking ? t_dev[0] : (negative ? -x : x);  // identical.py:70-71
    stamp(ev, 0);
    if (krum) {
      // dq_j = |h_j - byz(t)| in lane j of every wave.  The Byzantine row needs them in ascending order: every wave counts
      // its four rows' share of their stable rank, wave 1 adds the shares after the barrier.
      const double dq = honest ? rank_distance(attack_candidate_sq(a, w, c, t)) : kInf;
      if (k > 0) {
        PARTQ[wave * BM_MAX_ROWS + lane] = rank_share(dq, lane, wave * kRankChunk, h);
        __syncthreads();
      }
      stamp(ev, 1);
      if (wave == 0) {
        // honest row i (lane i): its h - 1 sorted honest distances merged with k copies of dq — `below` of them come
        // first (a binary search of the row: six dependent LDS reads) — and the `take` smallest added in ascending order
        // (loads in groups of eight at clamped indices, every one unconditional: they leave together and the additions
        // follow; a conditional load costs a branch and a full LDS latency per element)
        int below = 0, len = hm1;
#pragma unroll
        for (int it = 0; it < 6; ++it) {  // (h - 1 <= 63 values)
          const int half = len >> 1, mid = below + half;
          const double v = hs[mid < hm1 ? mid : last];
          const bool right = len > 0 && v < dq;
          below = right ? mid + 1 : below;
          len = right ? len - half - 1 : half;
        }
        stamp(ev, 2);
        // the merged sequence in its three stretches: b1 values of the row, c2 copies of dq, c3 more values of the row
        // from index b1 on (c3 > 0 only when b1 == below).  Every lane walks every stretch up to the longest one in the
        // wave and adds 0.0 beyond its own length: the distances are >= +0, so x + 0.0 is x bit for bit, and an element
        // costs a read at an immediate offset, a compare, a select and the addition (the one-loop form with its index
        // arithmetic per element: ~18 instructions, 4 000 cycles for 38 elements on a wave that issues one every 5-6).
        const int b1 = below < take ? below : take;
        const int c2 = (take - b1 < k) ? take - b1 : k;
        const int c3 = take - b1 - c2;
        const int most1 = longest(b1, take), most3 = longest(c3, take);
        double score = 0.0;
        for (int u0 = 0; u0 < most1; u0 += 8) {
          double g[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) g[q] = hs[u0 + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) score += (u0 + q < b1) ? g[q] : 0.0;
        }
        if (__builtin_amdgcn_ballot_w64(c2 > 0) != 0ull) {
          for (int u = 0; u < k; ++u) score += (u < c2) ? dq : 0.0;
        }
        const double* const hs3 = hs + b1;
        for (int u0 = 0; u0 < most3; u0 += 8) {  // (a group may run past the row's end: into the next row or the tables behind, unused)
          double g[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) g[q] = hs3[u0 + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) score += (u0 + q < c3) ? g[q] : 0.0;
        }
        if (honest) SC[lane] = score;
      } else if (wave == kByzWave && k > 0) {
        // the Byzantine row (all k are the same row): k - 1 zeros (its copies), then the dq in ascending order
        int place = 0;
#pragma unroll
        for (int q = 0; q < kSearchWaves; ++q) place += PARTQ[q * BM_MAX_ROWS + lane];
        if (honest) Q[place] = dq;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        stamp(ev, 2);
        const int zeros = (k - 1 < take) ? k - 1 : take;
        const int rest = take - zeros;
        double sb = 0.0;
#pragma unroll 8
        for (int u = 0; u < rest; ++u) sb += Q[u];  // (every lane the same sum: broadcast reads)
        if (lane < k) SC[h + lane] = sb;
      }
      stamp(ev, 3);
      __syncthreads();
      stamp(ev, 4);
      // stable argsort of the n scores (Python's sort, krum.py:62): wave c counts, for every row, how many of the rows
      // 4c .. 4c+3 come before it; wave 0 adds the sixteen counts
      if (lane < n) {
        const double si = SC[lane];
        int before = 0;
#pragma unroll
        for (int q = 0; q < kRankChunk; ++q) {
          const int j = wave * kRankChunk + q;
          const double sj = SC[j < n ? j : 0];
          before += (j < n && (sj < si || (sj == si && j < lane))) ? 1 : 0;
        }
        PART[wave * BM_MAX_ROWS + lane] = before;
      }
      __syncthreads();
      stamp(ev, 5);
    }
    if (wave == 0) {
      if (krum) {
        int rank = 0;
#pragma unroll
        for (int q = 0; q < kSearchWaves; ++q) rank += PART[q * BM_MAX_ROWS + (lane < n ? lane : 0)];
        selected = __builtin_amdgcn_ballot_w64(lane < n && rank < m);  // krum.py:78-80: the m best scores
        if (ranking) {
          // slot r names the row of rank r: the permutation is inverted through LDS (SC is free — every wave read its
          // scores before the last barrier), so that every entry of order_out is written once, by its own lane.  (The
          // ranks of the n rows are a permutation of 0 .. n-1 unless a score is NaN: a slot nobody claims reads 0.)
          int* const inv = reinterpret_cast<int*>(SC);
          inv[lane] = 0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (lane < n && rank < BM_MAX_ROWS) inv[rank] = lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          order_out[lane] = lane < n ? inv[lane] : 0;
        }
      }
      stamp(ev, 6);
      if (!ranking && (krum || ev == 0)) {
        // row sums of <u_i, u_j> over the selected honest j in index order, as attack_row_sum (search_core.h) forms them:
        // every column of the row's span adds its value or 0.0 (the loads of a group of eight leave together)
        // — in two halves of the span, two chains of additions in flight
        const unsigned long long on = selected & honest_rows;
        const int half = span >> 1;  // (a multiple of four)
        double lo = 0.0, hi = 0.0;
        for (int j0 = 0; j0 < half; j0 += 4) {
          double g[4], gh[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            g[q] = uu[j0 + q];
            gh[q] = uu[half + j0 + q];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            lo += ((on >> (j0 + q)) & 1ull) ? g[q] : 0.0;
            hi += ((on >> (half + j0 + q)) & 1ull) ? gh[q] : 0.0;
          }
        }
        row = lo + hi;
      }
      stamp(ev, 7);
      const bool mine = honest && ((selected >> lane) & 1ull);
      const double quad = butterfly_sum(mine ? row : 0.0), lin = butterfly_sum(mine ? w : 0.0);
      stamp(ev, 8);
      const int kb = __builtin_popcountll(h >= 64 ? 0ull : (selected >> h));
      const double y = attack_objective_value(quad, lin, kb, t, c, count);
      if (lane == 0 && !ranking) {
        out[1 + 2 * ev] = x;
        out[2 + 2 * ev] = y;
        Y[0] = y;
      }
      stamp(ev, 9);
    }
    __syncthreads();
    if (!ranking) cursor_report(&cur, Y[0]);
    stamp(ev, 10);
  }
  if (tid == 0 && !ranking) out[0] = cur.best_x;
}

// The cursor of the exploration kept in DEVICE memory, for the searches whose candidates are evaluated by d-sized kernels
// (median, trimmed mean, phocas, meamed, any rule): one lane takes the objective the last evaluation left on the device,
// moves the cursor and leaves the next candidate's factor on the device, where the evaluation kernels read it
// (bm_multi_fma3_bdev, bm_colwise_eval_tdev).  The host queues all the evaluations of a search without waiting for any.
__global__ void search_cursor_kernel(bm_search* __restrict__ state, const double* __restrict__ y, int negative, int last,
                                     double start, double delta, double ratio, double* __restrict__ t_out,
                                     double* __restrict__ out) {
  bm_search cur;
  if (y == nullptr) {
    cursor_begin(&cur, start, delta, ratio);
  } else {
    cur = *state;
    out[2 + 2 * cur.evaluations] = y[0];
    cursor_report(&cur, y[0]);
  }
  if (last) {
    out[0] = cur.best_x;
  } else {
    cursor_propose(&cur);
    out[1 + 2 * cur.evaluations] = cur.probe;
    t_out[0] = negative ? -cur.probe : cur.probe;  // identical.py:70-71
  }
  *state = cur;
}

}  // namespace bm

extern "C" int bm_search_device_next(void* state, const double* y, int negative, int last, double start, double delta,
                                     double ratio, double* t_out, double* out, void* stream) {
  using namespace bm;
  if (state == nullptr || out == nullptr || (!last && t_out == nullptr) || (last && y == nullptr)) return BM_EINVAL;
  if (y == nullptr && (!(start >= 0.0) || !(delta > 0.0) || !(ratio > 0.5 && ratio < 1.0))) return BM_EINVAL;
  hipLaunchKernelGGL(search_cursor_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream),
                     static_cast<bm_search*>(state), y, negative ? 1 : 0, last ? 1 : 0, start, delta, ratio, t_out, out);
  BM_LAUNCH_CHECK();
  return 0;
}

extern "C" int bm_attack_line_search_device(const double* ext, int h, int k, int f, int rule, int m, int evals,
                                            int negative, double* out, void* stream) {
  using namespace bm;
  const int n = h + k;
  if (ext == nullptr || out == nullptr || h < 1 || k < 0 || n > BM_MAX_ROWS || f < 0 || evals < 1) return BM_EINVAL;
  if (rule == BM_RULE_KRUM) {
    if (m <= 0) m = n - f - 2;
    if (m < 1 || m > n) return BM_EINVAL;
  } else if (rule != BM_RULE_AVERAGE) {
    return BM_EINVAL;  // Brute: the host form (bm_attack_line_search)
  }
  const size_t lds = search_lds_bytes(h);  // (39 honest rows: 35 KB; 64: 75 KB, behind the opt-in)
  const char* trace_env = getenv("BM_SEARCH_TRACE");
  const bool trace = trace_env != nullptr && trace_env[0] == '1';  // (the caller then holds 2 * kTraceSlots more doubles per candidate)
  auto kernel = trace ? attack_search_kernel<true> : attack_search_kernel<false>;
  const int rc = lds_opt_in(reinterpret_cast<const void*>(kernel), lds, 0);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(kernel, dim3(1), dim3(kSearchBlock), lds, static_cast<hipStream_t>(stream), ext, h, k, f, rule, m, evals,
                     negative ? 1 : 0, out, static_cast<const double*>(nullptr), static_cast<int32_t*>(nullptr), -1);
  BM_LAUNCH_CHECK();
  return 0;
}

// bm_attack_ranking (linesearch.cpp) on the device, the factor read from DEVICE memory: the ranking of
// honests + [avg + t att] * k from the (h+2)^2 scalars where bm_pairwise_sqdist left them.  One workgroup, the machinery
// of the search kernel above for one candidate (its set-up included: ~15 us); no copy, no synchronisation — the searches
// against Bulyan then keep their cursor on the device like the others.
extern "C" int bm_attack_ranking_device(const double* ext, int h, int k, int f, int mode, int m, const double* t_dev,
                                        int32_t* order_out, void* stream) {
  using namespace bm;
  const int n = h + k;
  if (ext == nullptr || order_out == nullptr || t_dev == nullptr || h < 1 || k < 1 || n > BM_MAX_ROWS || f < 0 ||
      (mode != BM_RANK_KRUM && mode != BM_RANK_BULYAN))
    return BM_EINVAL;
  if (m <= 0) m = n - f - 2;
  if (m < 1 || m > n) return BM_EINVAL;
  const int take = mode == BM_RANK_KRUM ? n - f - 1 : m;
  const size_t lds = search_lds_bytes(h);
  auto kernel = attack_search_kernel<false, true>;
  const int rc = lds_opt_in(reinterpret_cast<const void*>(kernel), lds, 0);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(kernel, dim3(1), dim3(kSearchBlock), lds, static_cast<hipStream_t>(stream), ext, h, k, f, (int)BM_RULE_KRUM,
                     m, 1, 0, static_cast<double*>(nullptr), t_dev, order_out, take);
  BM_LAUNCH_CHECK();
  return 0;
}
