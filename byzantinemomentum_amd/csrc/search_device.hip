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
//   * once, 16 waves: every honest row's distances to the other honest rows in ascending order (one bitonic network
//     per row and wave, as rank_body.h), and <u_i, u_j>;
//   * per candidate, ONE wave, a row per lane: lane i forms dq_i = |h_i - byz(t)|, counts how many of its sorted honest
//     distances lie below it and adds the `take` smallest of the merged sequence in ascending order (the additions of
//     rank_order() on the host: equal values in either order give the same sums); the Byzantine rows are all the same
//     row — k - 1 zeros, then the dq in ascending order (ranked across the lanes through v_readlane); stable argsort
//     of the n scores the same way, the selected set as a ballot; the objective from the selected set (row sums in
//     index order, then their sum).  No workgroup barrier and no conditional LDS load inside the search: the first
//     version (256 lanes, tables and flags in LDS, a barrier between the phases) took 216 us for the 16 candidates of
//     C3, every loop iteration waiting a full LDS latency behind a branch (profiles/r06_device_search.txt).
#include "bm_common.h"
#include "rank_body.h"
#include "search_core.h"

namespace bm {

// 16 waves for the set-up (every honest row's sort on a wave of its own, four rows at a time per wave); the candidates
// are evaluated by wave 0 alone, one row per lane, with the other waves gone: no workgroup barrier inside the search, the
// lanes exchange through v_readlane and through LDS in program order of the one wave.
constexpr int kSearchBlock = 1024;
constexpr int kSearchWaves = kSearchBlock / 64;
constexpr int kSearchRowsPerWave = (BM_MAX_ROWS + kSearchWaves - 1) / kSearchWaves;

__host__ __device__ inline int search_ld(int h) { return h | 1; }  // odd row length: lane i walks row i without bank conflicts
// LDS: UU[h][ld] (<u_i, u_j>), HS[h][ld] (row i's distances to the other honest rows, ascending), Q[h] (sorted dq)
__host__ __device__ inline size_t search_lds_bytes(int h) { return (size_t)(2 * h * search_ld(h) + h) * sizeof(double); }

__device__ __forceinline__ double lane_value(double v, int src) {  // lane `src` (wave-uniform) of v, through v_readlane
  const long long bits = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Among the first `count` lanes: how many hold a smaller value than this lane's, ties to the lower lane (a stable
// ascending rank; `count` wave-uniform).  v_readlane + two compares per lane visited, no LDS.
__device__ __forceinline__ int stable_rank(double v, int lane, int count) {
  int rank = 0;
  for (int j = 0; j < count; ++j) {
    const double o = lane_value(v, j);
    rank += (o < v || (o == v && j < lane)) ? 1 : 0;
  }
  return rank;
}

__global__ __launch_bounds__(kSearchBlock) void attack_search_kernel(const double* __restrict__ ext, int h, int k, int f,
                                                                     int rule, int m, int evals, int negative,
                                                                     double* __restrict__ out) {
  extern __shared__ double search_smem[];
  const int n = h + k, e = h + 2, tid = threadIdx.x, ld = search_ld(h);
  double* const UU = search_smem;
  double* const HS = UU + h * ld;
  double* const Q = HS + h * ld;
  const double c = ext[h * e + h + 1];
  const double kInf = __builtin_inf();

  // ---- once, by the whole workgroup: <u_i, u_j>, and every honest row's distances to the other honest rows in
  // ascending order (a 64-lane bitonic network per row: 21 exchange steps; up to four rows of a wave go through it
  // together so that their exchanges overlap)
  for (int p = tid; p < h * h; p += kSearchBlock) {
    const int i = p / h, j = p - i * h;
    UU[i * ld + j] = attack_uu(ext[i * e + h], ext[j * e + h], ext[i * e + j], i == j);
  }
  {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    double v[kSearchRowsPerWave];
#pragma unroll
    for (int r = 0; r < kSearchRowsPerWave; ++r) {
      const int i = wave + r * kSearchWaves;
      v[r] = (i < h && lane < h && lane != i) ? rank_distance(ext[i * e + lane]) : kInf;
    }
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
      for (int j = kk >> 1; j > 0; j >>= 1) {
        const bool keep_min = ((lane & kk) == 0) == ((lane & j) == 0);
#pragma unroll
        for (int r = 0; r < kSearchRowsPerWave; ++r) {
          const double o = __shfl_xor(v[r], j, 64);
          v[r] = keep_min ? __builtin_fmin(v[r], o) : __builtin_fmax(v[r], o);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kSearchRowsPerWave; ++r) {
      const int i = wave + r * kSearchWaves;
      if (i < h && lane < h - 1) HS[i * ld + lane] = v[r];
    }
  }
  __syncthreads();
  if (tid >= 64) return;  // (a barrier never waits for a wave that has ended; none follows anyway)

  // ---- the candidates, by wave 0: lane i < h is honest row i, lanes h .. n-1 stand for the Byzantine copies
  const int lane = tid;
  const bool honest = lane < h;
  const double a = honest ? ext[lane * e + h] : 0.0;
  const double w = honest ? attack_w(a, c, ext[lane * e + h + 1]) : 0.0;
  const double* const hs = HS + (honest ? lane : 0) * ld;
  const double* const uu = UU + (honest ? lane : 0) * ld;
  int take = n - f - 1;  // krum.py:59-60
  take = take > n - 1 ? n - 1 : take;
  take = take < 0 ? 0 : take;
  const int count = (rule == BM_RULE_KRUM) ? m : n;
  const int hm1 = h - 1, last = hm1 > 0 ? hm1 - 1 : 0;
  unsigned long long selected = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);  // Average: every row, at every candidate
  double row = 0.0;

  bm_search cur;
  cursor_begin(&cur, 0.0, 1.0, 0.8);  // the attack's call: tools.line_maximize(eval_factor, evals=evals)
  for (int ev = 0; ev < evals; ++ev) {  // (every lane runs the cursor: the same values in every lane)
    cursor_propose(&cur);
    const double x = cur.probe;
    const double t = negative ? -x : x;  // identical.py:70-71
    if (rule == BM_RULE_KRUM) {
      const double dq = honest ? rank_distance(attack_candidate_sq(a, w, c, t)) : kInf;
      // honest row i: its h - 1 sorted honest distances merged with k copies of dq — `below` of them come first.
      // (Loads in groups of eight at clamped indices, every one unconditional: they leave together and the arithmetic
      // follows; a conditional load costs a branch and a full LDS latency per element.)
      int below = 0;
      for (int j0 = 0; j0 < hm1; j0 += 8) {
        double g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = hs[(j0 + q < hm1) ? j0 + q : last];
#pragma unroll
        for (int q = 0; q < 8; ++q) below += (j0 + q < hm1 && g[q] < dq) ? 1 : 0;
      }
      double score = 0.0;
      for (int u0 = 0; u0 < take; u0 += 8) {
        double g[8];
        bool from_row[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int u = u0 + q;
          from_row[q] = (u < below) || (u >= below + k);
          int idx = (u < below) ? u : u - k;   // 0 <= idx < h - 1 whenever it is used: u < take <= h + k - 1
          idx = (from_row[q] && u < take) ? idx : 0;
          g[q] = hs[idx];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double v = from_row[q] ? g[q] : dq;
          score = (u0 + q < take) ? score + v : score;
        }
      }
      if (k > 0) {
        // a Byzantine row (all k are the same row): k - 1 zeros (its copies), then the dq in ascending order
        const int place = stable_rank(dq, lane, h);
        if (honest) Q[place] = dq;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int zeros = (k - 1 < take) ? k - 1 : take;
        const int rest = take - zeros;
        double sb = 0.0;
#pragma unroll 8
        for (int u = 0; u < rest; ++u) sb += Q[u];
        score = honest ? score : sb;
        __builtin_amdgcn_wave_barrier();  // (Q is rewritten by the next candidate)
      }
      const int rank = stable_rank(score, lane, n);  // stable argsort of the scores (Python's sort, krum.py:62)
      selected = __builtin_amdgcn_ballot_w64(lane < n && rank < m);  // krum.py:78-80: the m best scores
    }
    if (rule == BM_RULE_KRUM || ev == 0) {
      // row sums of <u_i, u_j> over the selected honest j, in index order
      row = 0.0;
      for (int j0 = 0; j0 < h; j0 += 8) {
        double g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = uu[(j0 + q < h) ? j0 + q : 0];
#pragma unroll
        for (int q = 0; q < 8; ++q) row = (j0 + q < h && ((selected >> (j0 + q)) & 1ull)) ? row + g[q] : row;
      }
    }
    double quad = 0.0, lin = 0.0;
    for (int i = 0; i < h; ++i) {
      if ((selected >> i) & 1ull) {  // (wave-uniform)
        quad += lane_value(row, i);
        lin += lane_value(w, i);
      }
    }
    const int kb = __builtin_popcountll(h >= 64 ? 0ull : (selected >> h));
    const double y = attack_objective_value(quad, lin, kb, t, c, count);
    if (lane == 0) {
      out[1 + 2 * ev] = x;
      out[2 + 2 * ev] = y;
    }
    cursor_report(&cur, y);
  }
  if (lane == 0) out[0] = cur.best_x;
}

}  // namespace bm

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
  const size_t lds = search_lds_bytes(h);
  const int rc = lds_opt_in(reinterpret_cast<const void*>(attack_search_kernel), lds, 0);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(attack_search_kernel, dim3(1), dim3(kSearchBlock), lds, static_cast<hipStream_t>(stream), ext, h,
                     k, f, rule, m, evals, negative ? 1 : 0, out);
  BM_LAUNCH_CHECK();
  return 0;
}
