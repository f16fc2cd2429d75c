// colwise_kernels.h — device code of the coordinate-wise rules (see colwise.hip for the contract).
#pragma once
#include "bm_common.h"

namespace bm {

constexpr int kColBlock = 256;

// Sum of the sorted ranks F .. N-F-1 in ascending order from 0, every register index static.
template <int N, int F>
__device__ __forceinline__ float trimmed_sum_static(const float (&x)[N]) {
  float t = 0.0f;
#pragma unroll
  for (int i = F; i < N - F; ++i) t += x[i];
  return t;
}

// The same for a wave-uniform run-time f: a scalar branch to the unrolled sum of that f.  (Written as
// `if (i >= f && i < N - f) t += x[i]` the compiler hoists the 2N rank predicates out of the column loop and
// keeps them in SGPRs: 22 of them spilled to VGPR lanes in the n = 25 burst kernel, which sits at its 128-VGPR
// limit.)  f outside 0 .. (N-1)/2 is refused by the host.
template <int N>
__device__ __forceinline__ float trimmed_sum(const float (&x)[N], int f) {
  float t = 0.0f;
  switch (f) {
#define BM_TRIM_CASE(F)                                             \
  case F:                                                           \
    if constexpr (2 * (F) < N) t = trimmed_sum_static<N, (F)>(x);   \
    break;
    BM_TRIM_CASE(0) BM_TRIM_CASE(1) BM_TRIM_CASE(2) BM_TRIM_CASE(3) BM_TRIM_CASE(4) BM_TRIM_CASE(5) BM_TRIM_CASE(6)
    BM_TRIM_CASE(7) BM_TRIM_CASE(8) BM_TRIM_CASE(9) BM_TRIM_CASE(10) BM_TRIM_CASE(11) BM_TRIM_CASE(12)
    BM_TRIM_CASE(13) BM_TRIM_CASE(14) BM_TRIM_CASE(15) BM_TRIM_CASE(16) BM_TRIM_CASE(17) BM_TRIM_CASE(18)
    BM_TRIM_CASE(19) BM_TRIM_CASE(20) BM_TRIM_CASE(21) BM_TRIM_CASE(22) BM_TRIM_CASE(23) BM_TRIM_CASE(24)
    BM_TRIM_CASE(25) BM_TRIM_CASE(26) BM_TRIM_CASE(27) BM_TRIM_CASE(28) BM_TRIM_CASE(29) BM_TRIM_CASE(30)
    BM_TRIM_CASE(31)
#undef BM_TRIM_CASE
    default:
      break;
  }
  return t;
}

// closest(): the sum of the m = N - F sorted values nearest the centre c (trmean.py:35-50).  In sorted order they form a
// window [s, s + m) with 0 <= s <= F: "value t is farther than value t + m" forces the window to start after t, and the
// LAST such t decides (with duplicated values the predicate is not monotone: take the maximum, not the count).  The
// window is summed in ascending order from s.  Everything on registers with static indices: rounds 2-5 wrote the
// sorted values to LDS and walked them with run-time indices (N ds_write + 2f + m dependent ds_read per column: at
// n = 51 the closest-to-centre rules ran at 0.59-0.60 of the HBM peak against 0.72 for the trimmed mean next to
// them); here the ranks F .. N-F-1 lie in every possible window and are added unconditionally, the 2F ranks at the
// ends are added as `x or 0` (adding +0 changes nothing: the sum has the bits of the walk from s).
template <int N, int F>
__device__ __forceinline__ float closest_sum_static(const float (&x)[N], float c) {
  constexpr int M = N - F;  // (M > F: the host refuses n < 2f + 1)
  int s = 0;
#pragma unroll
  for (int t = 0; t < F; ++t) {
    const float dl = __builtin_fabsf(x[t] - c);
    const float dh = __builtin_fabsf(x[t + M] - c);
    s = (dl > dh) ? (t + 1) : s;
  }
  float w = 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i < F)
      w += (i >= s) ? x[i] : 0.0f;
    else if (i >= M)
      w += (i < s + M) ? x[i] : 0.0f;
    else
      w += x[i];
  }
  return w;
}

// The same for a wave-uniform run-time f: a scalar branch to the instance of that f (see trimmed_sum).
template <int N>
__device__ __forceinline__ float closest_sum(const float (&x)[N], int f, float c) {
  float w = 0.0f;
  switch (f) {
#define BM_CLOSE_CASE(F)                                              \
  case F:                                                             \
    if constexpr (2 * (F) < N) w = closest_sum_static<N, (F)>(x, c);  \
    break;
    BM_CLOSE_CASE(0) BM_CLOSE_CASE(1) BM_CLOSE_CASE(2) BM_CLOSE_CASE(3) BM_CLOSE_CASE(4) BM_CLOSE_CASE(5) BM_CLOSE_CASE(6)
    BM_CLOSE_CASE(7) BM_CLOSE_CASE(8) BM_CLOSE_CASE(9) BM_CLOSE_CASE(10) BM_CLOSE_CASE(11) BM_CLOSE_CASE(12)
    BM_CLOSE_CASE(13) BM_CLOSE_CASE(14) BM_CLOSE_CASE(15) BM_CLOSE_CASE(16) BM_CLOSE_CASE(17) BM_CLOSE_CASE(18)
    BM_CLOSE_CASE(19) BM_CLOSE_CASE(20) BM_CLOSE_CASE(21) BM_CLOSE_CASE(22) BM_CLOSE_CASE(23) BM_CLOSE_CASE(24)
    BM_CLOSE_CASE(25) BM_CLOSE_CASE(26) BM_CLOSE_CASE(27) BM_CLOSE_CASE(28) BM_CLOSE_CASE(29) BM_CLOSE_CASE(30)
    BM_CLOSE_CASE(31)
#undef BM_CLOSE_CASE
    default:
      break;
  }
  return w;
}

// Per-column rule on N register-resident values (no LDS: the last argument is kept for the call sites' sake).
template <int N, int OP, int STRIDE = kColBlock>
__device__ __forceinline__ float column_rule(float (&x)[N], int f, float inv_keep, float* /*unused*/) {
  const float kNaN = __builtin_nanf("");
  const float kInf = __builtin_inff();
  // --- NaN scan (1 v_cmp per value, masks OR-ed on the scalar unit) ---
  bool has_nan = false;
#pragma unroll
  for (int i = 0; i < N; ++i) has_nan |= (x[i] != x[i]);

  if constexpr (OP == BM_OP_MEDIAN) {
    // torch.median: any NaN in the column -> NaN.  Other lanes are unaffected by a
    // NaN-polluted network in this lane, so no replacement pass is needed.
    sort_network<N>(x);
    const float med = x[(N - 1) / 2];
    return has_nan ? kNaN : med;
  } else {
    int nan_count = 0;
    if (__builtin_amdgcn_ballot_w64(has_nan) != 0ull) {  // wave-uniform, rarely taken
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const bool isn = (x[i] != x[i]);
        nan_count += isn ? 1 : 0;
        x[i] = isn ? kInf : x[i];  // NaN sorts last (torch.sort)
      }
    }
    sort_network<N>(x);
    // trimmed mean: ranks f .. N-f-1, summed in ascending order
    float tsum = 0.0f;
    if constexpr (OP == BM_OP_TRMEAN || OP == BM_OP_PHOCAS) tsum = trimmed_sum<N>(x, f);
    if constexpr (OP == BM_OP_TRMEAN) {
      const float r = div_small_int(tsum, (float)(N - 2 * f), inv_keep);
      return (nan_count > f) ? kNaN : r;
    } else {
      // closest(): mean of the m = N-f values nearest the centre (closest_sum above)
      float c;
      if constexpr (OP == BM_OP_PHOCAS) {
        c = div_small_int(tsum, (float)(N - 2 * f), 1.0f / (float)(N - 2 * f));
        if (nan_count > f) c = kNaN;
      } else {
        c = has_nan ? kNaN : x[(N - 1) / 2];
      }
      const int m = N - f;
      const float wsum = closest_sum<N>(x, f, c);
      const float r = div_small_int(wsum, (float)m, inv_keep);
      return (c != c || nan_count > f) ? kNaN : r;
    }
  }
}

// ABLATE_STORE (BM_COL_ABLATE=1, n = 25 only) keeps the arithmetic and drops the result store: the
// read-only rate of the same kernel.  Measured at n = 25, d = 11.2 M: 172 us without the store, 195 us
// with it — the 4 % of result bytes cost 11 % of the time.  It is not the store's place in the in-order
// vmcnt queue (scripts/probes/store_order_probe.hip: a store issued behind the next loads changes
// nothing, 189.8 vs 189.4 us on a bare 25-row stream) and not its cache policy (BM_RESULT_NT A/B): a
// write stream interleaved with 25 read streams simply costs about twice its bytes on this HBM system.
template <int N, int OP, int VEC, bool ABLATE_STORE = false>
__global__ __launch_bounds__(kColBlock) void colwise_kernel(RowTable rows, int64_t nvec, int tail,
                                                            int f, float inv_keep, int nt_result,
                                                            float* __restrict__ out) {
  float* const lds = nullptr;  // (no rule needs LDS any more, see closest_sum_static)
  // nvec * VEC * 4 < 2^32 (the host splits longer gradients): 32-bit byte offsets, saddr loads
  const uint32_t nv = (uint32_t)nvec;
  const uint32_t stride = gridDim.x * kColBlock;
  for (uint32_t v = blockIdx.x * kColBlock + threadIdx.x; v < nv; v += stride) {
    const uint32_t off = v * (uint32_t)(VEC * sizeof(float));
    float x[VEC][N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float t[VEC];
      load_stream_off<VEC>(rows.p[i], off, t);
#pragma unroll
      for (int c = 0; c < VEC; ++c) x[c][i] = t[c];
    }
    float r[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) r[c] = column_rule<N, OP>(x[c], f, inv_keep, lds);
    if constexpr (ABLATE_STORE) {
      if (r[0] == 1.2345678e-30f) store_stream_off<VEC>(out, off, r);
    } else if (nt_result) {  // wave-uniform; BM_RESULT_NT=0 selects the default cache policy (A/B runs)
      store_stream_off<VEC>(out, off, r);
    } else {
      store_result_off<VEC>(out, off, r);
    }
  }
  // the d % VEC trailing columns: one lane each, in the last workgroup (no second launch)
  if (VEC > 1 && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < tail) {
    const int64_t j = nvec * VEC + threadIdx.x;
    float x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = rows.p[i][j];
    out[j] = column_rule<N, OP>(x, f, inv_keep, lds);
  }
}

// ---------------------------------------------------------------------------------------------------
// Burst form (VEC = 4, N <= 25 — the closest-to-centre rules up to their own row limit, colwise_dispatch.h —, long gradients): the same loads and the same rule, but
// the results leave the CU in bursts that coincide across the chip.
//
// A result stream that trickles out between the reads of 256 CUs costs 10-14 % of the kernel for 3.8 % of
// its bytes (scripts/probes/rows_sweep_probe.hip).  Here ONE workgroup of 1024 lanes per CU (it declares
// all 160 KB of LDS, so exactly one fits) walks the columns interleaved with the other CUs — iteration `it`
// of workgroup b covers column groups (it * gridDim + b) * 1024 + lane, so at any time the chip reads one
// contiguous window of every row — stages the results of 10 iterations in LDS, meets at a barrier and
// writes them back to back.  Every CU started at the same time and has
// the same work, so the bursts line up across the chip without any global synchronisation.
// scripts/probes/two_phase_probe.hip on a bare 25-row stream: 194.3 us plain, 180.8 us in this form
// (175.6 us with no store at all); without the barrier 197.7 us, with contiguous instead of interleaved
// ownership 199.0 us, with bursts of 5 / 2 / 1 iterations 191.3 / 190.3 / 191.0 us.
// ---------------------------------------------------------------------------------------------------
constexpr int kBurstThreads = 1024;
constexpr int kBurstLdsBytes = 160 * 1024;

// Rows whose start addresses are congruent modulo 2 MB (one allocation per row: what a caller of the rules has) run
// this kernel 0-10 % slower than rows cut out of one allocation at a skewed stride, depending on the box (0 % on one,
// 5-6 % on two, 10 % on the round-4 driver box).  Three kernel-side remedies were measured on such rows, each in one
// process alternating with this kernel on the same data, and removed: the row order rotated per wave (no change), the
// four 256-byte quarters of a wave's 1 KB load spread 1 KB .. 256 KB apart (0-4 % slower), the row loads without the
// non-temporal hint (11 % slower, on the slab as well) — profiles/r05_b_col_placement_probe.txt,
// profiles/r05_e_col_load_policy_probe.txt.
template <int N, int OP, int VEC>
__global__ __launch_bounds__(kBurstThreads) void colwise_burst_kernel(RowTable rows, int64_t nvec, int tail, int f,
                                                                      float inv_keep, float* __restrict__ out) {
  using V = typename VecLoad<VEC>::T;
  constexpr int kSlots = kBurstLdsBytes / (kBurstThreads * VEC * (int)sizeof(float));
  __shared__ V stage[kSlots * kBurstThreads];
  const uint32_t nv = (uint32_t)nvec, tid = threadIdx.x;
  const uint32_t span = gridDim.x * kBurstThreads;  // column groups per iteration of the whole grid
  const uint32_t iters = (nv + span - 1) / span;
  const uint32_t first = blockIdx.x * kBurstThreads + tid;
  for (uint32_t p0 = 0; p0 < iters; p0 += kSlots) {
    const uint32_t p1 = (p0 + kSlots < iters) ? p0 + kSlots : iters;
    for (uint32_t it = p0; it < p1; ++it) {
      const uint32_t v = it * span + first;
      if (v < nv) {
        const uint32_t off = v * (uint32_t)(VEC * sizeof(float));
        float x[VEC][N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
          float t[VEC];
          load_stream_off<VEC>(rows.p[i], off, t);
#pragma unroll
          for (int c = 0; c < VEC; ++c) x[c][i] = t[c];
        }
        float r[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) r[c] = column_rule<N, OP>(x[c], f, inv_keep, nullptr);
        V packed;
        if constexpr (VEC == 1) {
          packed = r[0];
        } else {
#pragma unroll
          for (int c = 0; c < VEC; ++c) packed[c] = r[c];
        }
        stage[(it - p0) * kBurstThreads + tid] = packed;
      }
    }
    __syncthreads();  // not for the data (a lane reads back its own slots): it is what makes the stores a burst
    for (uint32_t it = p0; it < p1; ++it) {
      const uint32_t v = it * span + first;
      if (v < nv) {
        const uint32_t off = v * (uint32_t)(VEC * sizeof(float));
        const V packed = stage[(it - p0) * kBurstThreads + tid];
        __builtin_nontemporal_store(packed, reinterpret_cast<V*>(reinterpret_cast<char*>(out) + off));
      }
    }
  }
  // the d % VEC trailing columns: one lane each, in the last workgroup
  if (VEC > 1 && blockIdx.x == gridDim.x - 1 && (int)tid < tail) {
    const int64_t j = nvec * VEC + tid;
    float x[N];
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = rows.p[i][j];
    out[j] = column_rule<N, OP>(x, f, inv_keep, nullptr);
  }
}

// ---------------------------------------------------------------------------------------------------
// Evaluate-only kernels (search_eval.hip, bulyan.hip): a workgroup leaves ONE partial of the objective, at most kEvalMaxBlocks
// workgroups per launch, and the finish kernel adds the partials in a fixed order.
// ---------------------------------------------------------------------------------------------------
constexpr int kEvalMaxBlocks = 2048;

// out[0] = sum of the partials in a fixed order (1024 lanes: lane l adds l, l + 1024, ..., then the fixed tree of
// block_reduce_sum): the search waits for this number before it can propose the next candidate, so the 64-deep
// chain of dependent loads a single wave would walk (17.6 us measured) is worth removing
constexpr int kEvalFinishThreads = 1024;
template <int THREADS>  // (a template so that every translation unit that launches it may hold the definition)
__global__ __launch_bounds__(THREADS) void eval_finish_kernel(const double* __restrict__ partial, int nparts,
                                                                         double* __restrict__ out) {
  __shared__ double red[THREADS / 64];
  double tot = 0.0;
  for (int b = threadIdx.x; b < nparts; b += THREADS) tot += partial[b];
  const double r = block_reduce_sum<THREADS>(tot, red);
  if (threadIdx.x == 0) out[0] = r;
}

// Measured alternative, not kept: an LDS-DMA variant (per-wave private [N][1 KiB] LDS slot filled by
// global_load_lds_dwordx4, read back with ds_read_b128, next chunk's DMA in flight during the
// network, no barrier) ran at 229 / 244 us against 194 / 195 us for this kernel at n = 25,
// d = 11.2 M: LDS capacity limits it to 6 waves per CU and the sorting network no longer overlaps.

}  // namespace bm
