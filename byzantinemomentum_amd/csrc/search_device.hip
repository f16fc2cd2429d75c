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
//   * once: <u_i, u_j>, and every honest row's distances to the other honest rows in ascending order — one bitonic
//     network per row, the row then STAYS in the registers of its wave, one value per lane;
//   * per candidate: the wave of honest row i forms dq_i = |h_i - byz(t)|, counts with one ballot how many of the row's
//     sorted distances lie below it, lays the merged sequence (those, k copies of dq_i, the rest) out across its lanes
//     and adds the `take` smallest in ascending order through v_readlane (the additions of rank_order() on the host:
//     equal values in either order give the same sums) — fifteen waves, up to five rows each, the chains of a wave
//     interleaved; the sixteenth wave does the Byzantine row meanwhile (all k are the same row: k - 1 zeros, then the
//     dq in ascending order); the stable argsort of the n scores is counted by all sixteen waves, four rows each; wave 0
//     turns the selected set (a ballot) into the objective (row sums in index order, then their sum over the lanes in
//     the butterfly order of search_core.h).
//   Cost (profiles/r06_device_search.txt): 160 us for the 16 candidates of C3 (n = 51), ~10 us per candidate — what the
//   host form spends on its arithmetic (0.16 ms) without its copy and its stream synchronisation.  The reference's
//   semantics make every score a SEQUENTIAL sum (38 dependent fp64 additions per row and candidate) and a candidate is
//   ~2 000 dependent instructions end to end; a loop iteration around one dependent fp64 addition is 32 cycles (13 ns) on
//   this chip, 64 when a v_readlane pair feeds it (scripts/probes/one_workgroup_costs.hip), so a wave cannot beat a
//   5 GHz core on latency here.  Earlier forms: one wave working row-per-lane from tables in LDS 216 us (conditional
//   loads: a branch and a full LDS latency per element), 180 us with the loads batched; sixteen waves with the sorted
//   rows in registers 175 us; with the dq ranked by all waves and the objective's outer sums in butterfly order 160 us.
#include "bm_common.h"
#include "rank_body.h"
#include "search_core.h"

namespace bm {

// One workgroup of 16 waves.  Waves 0 .. 14 own the honest rows (row i on wave i % 15, the row's sorted distances one per
// lane, in registers for the whole search), wave 15 owns the Byzantine row; wave 0 also turns the scores into the
// objective.  Three workgroup barriers per candidate.
constexpr int kSearchBlock = 1024;
constexpr int kSearchWaves = kSearchBlock / 64;
constexpr int kRowWaves = kSearchWaves - 1;
constexpr int kRowsPerWave = (BM_MAX_ROWS + kRowWaves - 1) / kRowWaves;       // 5: every h <= n <= 64 (the step itself stops at h = 62: its distance pass takes 64 rows)
constexpr int kRankChunk = BM_MAX_ROWS / kSearchWaves;                       // rows whose scores one wave compares with everybody's

__host__ __device__ inline int search_ld(int h) { return h | 1; }  // odd row length: lane i walks row i without bank conflicts
// LDS: UU[h][ld] (<u_i, u_j>), SC[64] (scores), Q[64] (sorted dq), Y[2] (objective), PART / PARTQ[16][64] (partial ranks
// of the scores / of the dq, int)
__host__ __device__ inline size_t search_lds_bytes(int h) {
  return (size_t)(h * search_ld(h) + 2 * BM_MAX_ROWS + 2) * sizeof(double) + (size_t)2 * kSearchWaves * BM_MAX_ROWS * sizeof(int);
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

// The sum of one value per lane over all 64 lanes in a FIXED order every lane can follow at once: v <- v + (v of lane
// ^ 1), then ^ 2, ^ 4, ... ^ 32 (fp64 addition commutes, so both partners of an exchange form the same sum and all lanes
// end with the same bits).  The host form adds in the same order (linesearch.cpp, butterfly_sum).
__device__ __forceinline__ double butterfly_sum(double v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) v = v + __shfl_xor(v, s, 64);
  return v;
}

// s[r] = ((0 + val[r]@lane 0) + val[r]@lane 1) + ... over the first `take` lanes, for the NR rows of a wave together
// (NR independent chains of fp64 additions: the chains hide one another's latency).
template <int NR>
__device__ __forceinline__ void fold_lanes(const double (&val)[kRowsPerWave], int take, double (&s)[kRowsPerWave]) {
#pragma unroll
  for (int r = 0; r < NR; ++r) s[r] = 0.0;
  for (int u = 0; u < take; ++u) {
#pragma unroll
    for (int r = 0; r < NR; ++r) s[r] += lane_value(val[r], u);
  }
}

__global__ __launch_bounds__(kSearchBlock) void attack_search_kernel(const double* __restrict__ ext, int h, int k, int f,
                                                                     int rule, int m, int evals, int negative,
                                                                     double* __restrict__ out) {
  extern __shared__ double search_smem[];
  const int n = h + k, e = h + 2, tid = threadIdx.x, ld = search_ld(h);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  double* const UU = search_smem;
  double* const SC = UU + h * ld;
  double* const Q = SC + BM_MAX_ROWS;
  double* const Y = Q + BM_MAX_ROWS;
  int* const PART = reinterpret_cast<int*>(Y + 2);
  int* const PARTQ = PART + kSearchWaves * BM_MAX_ROWS;
  const double c = ext[h * e + h + 1];
  const double kInf = __builtin_inf();
  const bool krum = rule == BM_RULE_KRUM;

  // ---- once: <u_i, u_j> (everybody); on the row waves every honest row's distances to the other honest rows in
  // ascending order, one per lane (a 64-lane bitonic network per row, the rows of a wave going through it together),
  // and the same values k lanes further up (the part of the merged sequence behind the k copies of the candidate)
  for (int p = tid; p < h * h; p += kSearchBlock) {
    const int i = p / h, j = p - i * h;
    UU[i * ld + j] = attack_uu(ext[i * e + h], ext[j * e + h], ext[i * e + j], i == j);
  }
  double sorted[kRowsPerWave], shifted[kRowsPerWave], row_a[kRowsPerWave], row_w[kRowsPerWave];
  int my_rows = 0;
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) {
    const int i = wave + r * kRowWaves;
    const bool mine = wave < kRowWaves && i < h;  // (wave-uniform)
    my_rows += mine ? 1 : 0;
    sorted[r] = (mine && lane < h && lane != i) ? rank_distance(ext[i * e + lane]) : kInf;
    row_a[r] = mine ? ext[i * e + h] : 0.0;
    row_w[r] = mine ? attack_w(row_a[r], c, ext[i * e + h + 1]) : 0.0;
  }
  if (wave < kRowWaves && krum) {
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
      for (int j = kk >> 1; j > 0; j >>= 1) {
        const bool keep_min = ((lane & kk) == 0) == ((lane & j) == 0);
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
          const double o = __shfl_xor(sorted[r], j, 64);
          sorted[r] = keep_min ? __builtin_fmin(sorted[r], o) : __builtin_fmax(sorted[r], o);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kRowsPerWave; ++r) shifted[r] = __shfl(sorted[r], lane >= k ? lane - k : 0, 64);
  // wave 15: lane j is honest row j as the Byzantine row sees it; wave 0: lane i is honest row i in the objective
  const bool honest = lane < h;
  const double a = honest ? ext[lane * e + h] : 0.0;
  const double w = honest ? attack_w(a, c, ext[lane * e + h + 1]) : 0.0;
  const double* const uu = UU + (honest ? lane : 0) * ld;
  int take = n - f - 1;  // krum.py:59-60
  take = take > n - 1 ? n - 1 : take;
  take = take < 0 ? 0 : take;
  const int count = krum ? m : n;
  unsigned long long selected = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);  // Average: every row, at every candidate
  double row = 0.0;
  __syncthreads();

  bm_search cur;
  cursor_begin(&cur, 0.0, 1.0, 0.8);  // the attack's call: tools.line_maximize(eval_factor, evals=evals)
  for (int ev = 0; ev < evals; ++ev) {  // (every lane runs the cursor: the same values everywhere)
    cursor_propose(&cur);
    const double x = cur.probe;
    const double t = negative ? -x : x;  // identical.py:70-71
    if (krum) {
      // the Byzantine row needs the dq_j = |h_j - byz(t)| in ascending order: every wave forms them (lane j: row j) and
      // counts its four rows' share of their stable rank; the sixteenth wave adds the shares after the barrier
      const double dq_lane = honest ? rank_distance(attack_candidate_sq(a, w, c, t)) : kInf;
      if (k > 0) {
        PARTQ[wave * BM_MAX_ROWS + lane] = rank_share(dq_lane, lane, wave * kRankChunk, h);
        __syncthreads();
      }
      if (wave < kRowWaves) {
        // honest row i: its h - 1 sorted honest distances merged with k copies of dq = |h_i - byz(t)| — `below` of them
        // come first — as one value per lane, then the `take` smallest added in ascending order
        double val[kRowsPerWave], s[kRowsPerWave];
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
          const double dq = rank_distance(attack_candidate_sq(row_a[r], row_w[r], c, t));
          const int below = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < h - 1 && sorted[r] < dq));
          val[r] = (lane < below) ? sorted[r] : ((lane < below + k) ? dq : shifted[r]);
        }
        switch (my_rows) {
          case 1: fold_lanes<1>(val, take, s); break;
          case 2: fold_lanes<2>(val, take, s); break;
          case 3: fold_lanes<3>(val, take, s); break;
          case 4: fold_lanes<4>(val, take, s); break;
          case 5: fold_lanes<5>(val, take, s); break;
          default: break;
        }
        if (lane == 0) {
#pragma unroll
          for (int r = 0; r < kRowsPerWave; ++r)
            if (r < my_rows) SC[wave + r * kRowWaves] = s[r];
        }
      } else if (k > 0) {
        // the Byzantine row (all k are the same row): k - 1 zeros (its copies), then the dq in ascending order
        int place = 0;
#pragma unroll
        for (int q = 0; q < kSearchWaves; ++q) place += PARTQ[q * BM_MAX_ROWS + lane];
        if (honest) Q[place] = dq_lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double ascending = honest ? Q[lane] : 0.0;
        const int zeros = (k - 1 < take) ? k - 1 : take;
        const int rest = take - zeros;
        double sb = 0.0;
        for (int u = 0; u < rest; ++u) sb += lane_value(ascending, u);
        if (lane < k) SC[h + lane] = sb;
      }
      __syncthreads();
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
    }
    if (wave == 0) {
      if (krum) {
        int rank = 0;
#pragma unroll
        for (int q = 0; q < kSearchWaves; ++q) rank += PART[q * BM_MAX_ROWS + (lane < n ? lane : 0)];
        selected = __builtin_amdgcn_ballot_w64(lane < n && rank < m);  // krum.py:78-80: the m best scores
      }
      if (krum || ev == 0) {
        // row sums of <u_i, u_j> over the selected honest j, in index order (loads in groups of eight at clamped
        // indices, every one unconditional: they leave together and the arithmetic follows)
        row = 0.0;
        for (int j0 = 0; j0 < h; j0 += 8) {
          double g[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) g[q] = uu[(j0 + q < h) ? j0 + q : 0];
#pragma unroll
          for (int q = 0; q < 8; ++q) row = (j0 + q < h && ((selected >> (j0 + q)) & 1ull)) ? row + g[q] : row;
        }
      }
      const bool on = honest && ((selected >> lane) & 1ull);
      const double quad = butterfly_sum(on ? row : 0.0), lin = butterfly_sum(on ? w : 0.0);
      const int kb = __builtin_popcountll(h >= 64 ? 0ull : (selected >> h));
      const double y = attack_objective_value(quad, lin, kb, t, c, count);
      if (lane == 0) {
        out[1 + 2 * ev] = x;
        out[2 + 2 * ev] = y;
        Y[0] = y;
      }
    }
    __syncthreads();
    cursor_report(&cur, Y[0]);
  }
  if (tid == 0) out[0] = cur.best_x;
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
  const size_t lds = search_lds_bytes(h);  // (at most 62 * 63 * 8 + 5 KB = 36 KB)
  const int rc = lds_opt_in(reinterpret_cast<const void*>(attack_search_kernel), lds, 0);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(attack_search_kernel, dim3(1), dim3(kSearchBlock), lds, static_cast<hipStream_t>(stream), ext, h,
                     k, f, rule, m, evals, negative ? 1 : 0, out);
  BM_LAUNCH_CHECK();
  return 0;
}
