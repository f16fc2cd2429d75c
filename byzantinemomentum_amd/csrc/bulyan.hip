// bulyan.hip — Bulyan pass 2 and Aksel pass 1: column kernels that consume a device-resident
// ranking, so the whole rule runs without a host synchronisation.
//
// Bulyan pass 2 replaces aggregators/bulyan.py:64-84.  The reference's score update is dead
// code (its guard `gid == gid_prune` can never hold for scores[1:], bulyan.py:74-76), so the
// scores are static: with order = stable argsort of the initial scores,
//     selected[i] = (0 + G[order[i]] + G[order[i+1]] + ...)/c_i ,  c_i = min(m, m_max - i)
// and the output is the coordinate-wise mean of the beta = theta-2f values of `selected`
// closest to their lower median.  One pass: read the m_max ranked rows once (4*d*m_max bytes),
// write d floats; the theta x d `selected` tensor never exists.
//
// Aksel pass 1 replaces aggregators/aksel.py:35-41: coordinate-wise lower median fused with the
// n per-row squared distances to it (one read of the n rows instead of n+2).
#include "colwise_kernels.h"

namespace bm {

constexpr int kBulBlock = 256;

// Bulyan's second pass on the columns a lane holds (W of them: VEC in the body, 1 for the d % VEC trailing columns), given
// their MMAX ranked values: shared by the kernel that writes the result and the one that only evaluates it.
template <int N, int F, typename X, typename R>
__device__ __forceinline__ void bulyan_columns(X& x, R& r, int short_window) {
  constexpr int MMAX = N - F - 2;
  constexpr int THETA = N - 2 * F - 2;
  constexpr int BETA = THETA - 2 * F;
  const float kNaN = __builtin_nanf("");
  {
    constexpr int W = (int)(sizeof(r) / sizeof(float));
#pragma unroll
    for (int c = 0; c < W; ++c) {
      // selected[i]: forward sequential sum from rank i to m_max-1, exact division by the count
      float sel[THETA];
      bool all_finite = true;
#pragma unroll
      for (int i = 0; i < THETA; ++i) {
        float s = 0.0f;
#pragma unroll
        for (int t = i; t < MMAX; ++t) s += x[c][t];
        const float cnt = (float)(MMAX - i);
        sel[i] = div_small_int(s, cnt, 1.0f / cnt);
        all_finite &= (__builtin_fabsf(sel[i]) < __builtin_inff());  // (false for NaN as well)
      }
      constexpr int MED = (THETA - 1) / 2;
      // beta closest to the median: window [s, s+BETA) of the sorted values, s = 1 + the last t with
      // |sel[t] - med| > |sel[t+BETA] - med|.  With FINITE values only the t whose window straddles the median can decide
      // anything that shows in the result: for t + BETA <= MED both values lie at or below the median, the test is
      // sel[t] < sel[t+BETA], and when such a t is the last one to fire every value from t+1 up to the median EQUALS the
      // median — the window it selects and the default window [MED-BETA+1, MED] then hold the same BETA values; for
      // t >= MED the test never fires.  So the short form looks at BETA-1 positions and sums over 2 BETA - 1 instead of
      // THETA - BETA and THETA (n = 25, f = 5: 2 and 5 instead of 10 and 13): same bits, a fifth of the kernel's VALU work
      // less.  It needs the sorted values around the median ONLY, so its sorting network — its own copy, inside the branch
      // — is pruned by dead-code elimination to a selection network (n = 51, f = 12: beta = 1, the result IS the median of
      // the 25 values; round 6, until then the full sorter ran in front of both forms).  A wave that holds a column with
      // a non-finite value takes the long form for all its columns (wave-uniform branch).
      if (short_window != 0 && __builtin_amdgcn_ballot_w64(!all_finite) == 0ull) {
        sort_network<THETA>(sel);
        const float med = sel[MED];
        constexpr int LO = (MED - BETA + 1 > 0) ? MED - BETA + 1 : 0;               // first t whose window reaches the median
        constexpr int TEND = (MED < THETA - BETA) ? MED : THETA - BETA;             // t >= MED never fires
        constexpr int IEND = (MED + BETA < THETA) ? MED + BETA : THETA;
        int s0 = LO;
#pragma unroll
        for (int t = LO; t < TEND; ++t) {
          const float dl = __builtin_fabsf(sel[t] - med);
          const float dh = __builtin_fabsf(sel[t + BETA] - med);
          s0 = (dl > dh) ? (t + 1) : s0;
        }
        float w = 0.0f;
#pragma unroll
        for (int i = LO; i < IEND; ++i) w += (i >= s0 && i < s0 + BETA) ? sel[i] : 0.0f;
        r[c] = div_small_int(w, (float)BETA, 1.0f / (float)BETA);
      } else {
        bool has_nan = false;
#pragma unroll
        for (int i = 0; i < THETA; ++i) has_nan |= (sel[i] != sel[i]);
        sort_network<THETA>(sel);
        const float med = sel[MED];
        int s0 = 0;
#pragma unroll
        for (int t = 0; t < THETA - BETA; ++t) {
          const float dl = __builtin_fabsf(sel[t] - med);
          const float dh = __builtin_fabsf(sel[t + BETA] - med);
          s0 = (dl > dh) ? (t + 1) : s0;
        }
        float w = 0.0f;
#pragma unroll
        for (int i = 0; i < THETA; ++i) w += (i >= s0 && i < s0 + BETA) ? sel[i] : 0.0f;
        const float res = div_small_int(w, (float)BETA, 1.0f / (float)BETA);
        r[c] = has_nan ? kNaN : res;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Register-resident Bulyan pass 2 for a compile-time (n, f) and the default m = m_max.
// ---------------------------------------------------------------------------
//
// (The chip-wide store-burst form that pays for the column kernels and the selected mean does not pay here: measured
// in round 3 with identical checksums, 153.0 -> 154.1 us at n = 25, 357.7 -> 359.1 us at n = 51, 87.2 -> 85.8 us at
// n = 15, profiles/r03_c_bulyan_pass2_burst_ab.txt.  It was removed.)
template <int N, int F, int VEC>
__global__ __launch_bounds__(kBulBlock) void bulyan_pass2_kernel(
    RowTable rows, const int32_t* __restrict__ order, int64_t nvec, int nt_result, float* __restrict__ out,
    int short_window, int tail) {
  constexpr int MMAX = N - F - 2;
  constexpr int THETA = N - 2 * F - 2;
  constexpr int BETA = THETA - 2 * F;
  static_assert(BETA >= 1, "bulyan needs n >= 4f+3");
  // The m_max ranked row pointers live in SGPRs (loads then use the saddr form with one 32-bit byte offset
  // per lane, like the column kernels).  Up to 25 of them are fetched with scalar loads only: `order` is
  // uniform, and the row table — the first kernel argument, passed by value — is indexed in the kernarg
  // segment itself.  A workgroup handles one column group per lane, so this prologue runs once per 256
  // results: through LDS, a barrier and readfirstlane it costs 5 % of the kernel at n = 25 (163.5 -> 155.4 us,
  // profiles/r02_g_bulyan_pass2_prologue.txt).  Above 25 pointers the scalar form runs out of SGPRs (16-60
  // spilled, n = 51: 365 -> 390 us), so the larger instances keep the LDS form.
  const float* ranked[MMAX];
  if constexpr (MMAX <= 25) {
    typedef const float* __attribute__((address_space(4))) const* KargTable;
    const KargTable karg = (KargTable)__builtin_amdgcn_kernarg_segment_ptr();
    // (the ranking: ONE vector load per wave — lane t fetches order[t] — then a v_readlane per pointer, see load_index_coherent)
    const int mine = (int)(threadIdx.x & 63) < MMAX ? load_index_coherent(order + (threadIdx.x & 63)) : 0;
#pragma unroll
    for (int t = 0; t < MMAX; ++t) ranked[t] = (const float*)karg[__builtin_amdgcn_readlane(mine, t)];
  } else {
    __shared__ const float* ranked_lds[MMAX];
    if (threadIdx.x < MMAX) ranked_lds[threadIdx.x] = rows.p[load_index_coherent(order + threadIdx.x)];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MMAX; ++t) {
      const uint64_t p = reinterpret_cast<uint64_t>(ranked_lds[t]);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
      const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
      ranked[t] = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
    }
  }
  const float kNaN = __builtin_nanf("");
  const uint32_t nv = (uint32_t)nvec;  // nvec * VEC * 4 < 2^32: the host splits longer gradients
  auto load_group = [&](uint32_t off, float (&x)[VEC][MMAX]) {
#pragma unroll
    for (int t = 0; t < MMAX; ++t) {
      float tmp[VEC];
      load_stream_off<VEC>(ranked[t], off, tmp);
#pragma unroll
      for (int c = 0; c < VEC; ++c) x[c][t] = tmp[c];
    }
  };
  auto rule = [&](auto& x, auto& r) { bulyan_columns<N, F>(x, r, short_window); };
  // the d % VEC trailing columns: one lane each, in the last workgroup (no second launch: 4.3 us of a C4 aggregation),
  // before the body so that nothing of it stays live across the main loop
  if (VEC > 1 && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < tail) {
    const int64_t j = nvec * VEC + threadIdx.x;
    float x[1][MMAX], r[1];
#pragma unroll
    for (int t = 0; t < MMAX; ++t) x[0][t] = ranked[t][j];
    rule(x, r);
    out[j] = r[0];
  }
  {
    const uint32_t nblk = (nv + kBulBlock - 1) / kBulBlock;
    for (uint32_t b = blockIdx.x; b < nblk; b += gridDim.x) {
      const uint32_t v = b * kBulBlock + threadIdx.x;
      if (v < nv) {
        const uint32_t off = v * (uint32_t)(VEC * sizeof(float));
        float x[VEC][MMAX], r[VEC];
        load_group(off, x);
        rule(x, r);
        store_result_policy<VEC>(reinterpret_cast<float*>(reinterpret_cast<char*>(out) + off), r, nt_result);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Bulyan pass 2, evaluate only: ONE candidate of the attacks' factor search (attacks/identical.py:67-77) against Bulyan.
// ---------------------------------------------------------------------------
// The search ranks the stack honests + [avg + t * dir] * k from scalars (bm_attack_ranking) and needs, per candidate,
// | pass2(ranked rows) - avg |^2.  The per-evaluation form writes the candidate vector (2 rows read, 1 written), runs
// bm_bulyan_pass2 (m_max rows read — the copies of the candidate from cache —, 1 written) and bm_sqdist2 (2 read): four
// launches.  Here the candidate exists in registers only — fma(t, dir, 1 * avg), the bits bm_multi_fma3 writes —, the
// ranks that hold one of its copies take it there (a wave-uniform bit per rank; their table entries point at avg, so
// that the load of such a rank is a harmless hit), the rule is the code of the kernel above (bulyan_columns) and the
// objective is accumulated like bm_colwise_eval / bm_sqdist2 do (fp32 over 16 column groups per lane, fp64 beyond, one
// partial per workgroup, eval_finish_kernel): the honest ranked rows + avg + dir read, NOTHING written, two launches.
struct Pass2Candidate {
  const float* avg;
  const float* dir;
  const double* t_dev;  // the factor in device memory (the device cursor's), or null: t_host
  float t_host;
  int h;                // table entries h .. n-1 are copies of the candidate
};

template <int N, int F, int VEC>
__global__ __launch_bounds__(kBulBlock) void bulyan_pass2_eval_kernel(RowTable rows, const int32_t* __restrict__ order,
                                                                      int64_t nvec, int short_window, Pass2Candidate cd,
                                                                      double* __restrict__ partial) {
  constexpr int MMAX = N - F - 2;
  static_assert(N - 4 * F - 2 >= 1 && MMAX <= 64, "bulyan needs n >= 4f+3");
  __shared__ const float* ranked_lds[MMAX];
  __shared__ unsigned long long copies_lds;
  __shared__ double red[kBulBlock / 64];
  // (the prologue of the kernel above in its LDS form for every size: it runs once per workgroup, and the workgroups of
  //  this kernel walk many column groups)
  if (threadIdx.x < 64) {  // wave 0 (MMAX <= 64)
    const int idx = (int)threadIdx.x < MMAX ? load_index_coherent(order + threadIdx.x) : 0;
    if ((int)threadIdx.x < MMAX) ranked_lds[threadIdx.x] = rows.p[idx];
    const unsigned long long copies = __builtin_amdgcn_ballot_w64((int)threadIdx.x < MMAX && idx >= cd.h);
    if (threadIdx.x == 0) copies_lds = copies;
  }
  __syncthreads();
  const float* ranked[MMAX];
#pragma unroll
  for (int t = 0; t < MMAX; ++t) {
    const uint64_t p = reinterpret_cast<uint64_t>(ranked_lds[t]);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    ranked[t] = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
  }
  const uint64_t cbits = copies_lds;
  const uint32_t copies_lo = __builtin_amdgcn_readfirstlane((uint32_t)cbits);
  const uint32_t copies_hi = __builtin_amdgcn_readfirstlane((uint32_t)(cbits >> 32));
  const float tf = cd.t_dev != nullptr ? (float)cd.t_dev[0] : cd.t_host;
  float acc = 0.0f;
  double wide = 0.0;
  int since = 0;
  // nvec * VEC * 4 < 2^32 (the host refuses longer vectors): 32-bit byte offsets, saddr loads, as in the kernel above
  const uint32_t nv = (uint32_t)nvec, stride = gridDim.x * kBulBlock;
  for (uint32_t v = blockIdx.x * kBulBlock + threadIdx.x; v < nv; v += stride) {
    const uint32_t off = v * (uint32_t)(VEC * sizeof(float));
    float x[VEC][MMAX], r[VEC], a[VEC], dr[VEC], cand[VEC];
    load_stream_off<VEC>(cd.avg, off, a);
    load_stream_off<VEC>(cd.dir, off, dr);
#pragma unroll
    for (int t = 0; t < MMAX; ++t) {
      float tmp[VEC];
      load_stream_off<VEC>(ranked[t], off, tmp);
#pragma unroll
      for (int c = 0; c < VEC; ++c) x[c][t] = tmp[c];
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) cand[c] = __builtin_fmaf(tf, dr[c], 1.0f * a[c]);  // bm_multi_fma3(out, avg, dir, 1, t): same bits
#pragma unroll
    for (int t = 0; t < MMAX; ++t) {
      const bool copy = (((t < 32 ? copies_lo : copies_hi) >> (t & 31)) & 1u) != 0u;  // wave-uniform
#pragma unroll
      for (int c = 0; c < VEC; ++c) x[c][t] = copy ? cand[c] : x[c][t];
    }
    bulyan_columns<N, F>(x, r, short_window);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      const float df = r[c] - a[c];  // aggregated.sub_(grad_avg) (identical.py:75)
      acc = __builtin_fmaf(df, df, acc);
    }
    if (++since == 16) {
      wide += (double)acc;
      acc = 0.0f;
      since = 0;
    }
  }
  const double tot = block_reduce_sum<kBulBlock>(wide + (double)acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

template <int N, int F>
static int launch_bulyan_eval(const float* const* honests, int h, const int32_t* order, int64_t d, const float* avg,
                              const float* dir, float t, const double* t_dev, double* out, double* partial, hipStream_t s) {
  constexpr int MMAX = N - F - 2;
  constexpr int kMaxVec = (MMAX <= 20) ? 4 : (MMAX <= 44 ? 2 : 1);
  RowTable tab{};
  for (int i = 0; i < N; ++i) tab.p[i] = i < h ? honests[i] : avg;
  const void* more[2] = {avg, dir};
  int vec = common_vec_width(reinterpret_cast<const void* const*>(honests), h, nullptr);
  const int vec2 = common_vec_width(more, 2, nullptr);
  if (vec2 < vec) vec = vec2;
  if (vec > kMaxVec) vec = kMaxVec;
  const Pass2Candidate cd{avg, dir, t_dev, t, h};
  int nparts = 0;
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    const int grid = stream_grid(nvec, kBulBlock, kEvalMaxBlocks);
    auto kern = vec == 4 ? bulyan_pass2_eval_kernel<N, F, (kMaxVec >= 4 ? 4 : 2)> : bulyan_pass2_eval_kernel<N, F, (kMaxVec >= 2 ? 2 : 1)>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBulBlock), 0, s, tab, order, nvec, tuning().bulyan_short, cd, partial);
    BM_LAUNCH_CHECK();
    nparts = grid;
    body = nvec * vec;
  }
  if (body < d) {
    RowTable tail{};
    for (int i = 0; i < N; ++i) tail.p[i] = (i < h ? honests[i] : avg) + body;
    const Pass2Candidate ct{avg + body, dir + body, t_dev, t, h};
    const int64_t rest = d - body;
    const int grid = (body == 0) ? stream_grid(rest, kBulBlock, kEvalMaxBlocks) : 1;
    hipLaunchKernelGGL((bulyan_pass2_eval_kernel<N, F, 1>), dim3(grid), dim3(kBulBlock), 0, s, tail, order, rest,
                       tuning().bulyan_short, ct, partial + nparts);
    BM_LAUNCH_CHECK();
    nparts += grid;
  }
  hipLaunchKernelGGL(eval_finish_kernel<kEvalFinishThreads>, dim3(1), dim3(kEvalFinishThreads), 0, s, partial, nparts, out);
  BM_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// Generic Bulyan pass 2 (any n, f, m): per-lane arrays live in LDS, column-major so that lane l
// always hits bank l.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBulBlock) void bulyan_pass2_generic_kernel(
    RowTable rows, const int32_t* __restrict__ order, int n, int f, int m, int64_t d,
    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int m_max = n - f - 2;
  const int theta = n - 2 * f - 2;
  const int beta = theta - 2 * f;
  __shared__ const float* ranked[BM_MAX_ROWS];
  if ((int)threadIdx.x < m_max) ranked[threadIdx.x] = rows.p[load_index_coherent(order + threadIdx.x)];
  __syncthreads();
  float* xs = smem + threadIdx.x;                       // [m_max][kBulBlock]
  float* sel = smem + m_max * kBulBlock + threadIdx.x;  // [theta][kBulBlock]
  const float kNaN = __builtin_nanf("");
  const int64_t stride = (int64_t)gridDim.x * kBulBlock;
  for (int64_t j = (int64_t)blockIdx.x * kBulBlock + threadIdx.x; j < d; j += stride) {
    for (int t = 0; t < m_max; ++t) xs[t * kBulBlock] = __builtin_nontemporal_load(ranked[t] + j);
    bool has_nan = false;
    for (int i = 0; i < theta; ++i) {
      int cnt = m_max - i;
      if (cnt > m) cnt = m;
      float s = 0.0f;
      for (int t = 0; t < cnt; ++t) s += xs[(i + t) * kBulBlock];
      float val = s / (float)cnt;
      has_nan |= (val != val);
      if (val != val) val = __builtin_inff();
      // insertion sort (ascending)
      int p = i;
      while (p > 0 && sel[(p - 1) * kBulBlock] > val) {
        sel[p * kBulBlock] = sel[(p - 1) * kBulBlock];
        --p;
      }
      sel[p * kBulBlock] = val;
    }
    const float med = sel[((theta - 1) / 2) * kBulBlock];
    int lo = 0, hi = theta - 1;
    for (int drop = 0; drop < theta - beta; ++drop) {
      const float dl = __builtin_fabsf(sel[lo * kBulBlock] - med);
      const float dh = __builtin_fabsf(sel[hi * kBulBlock] - med);
      if (dl > dh) ++lo; else --hi;
    }
    float w = 0.0f;
    for (int i = lo; i <= hi; ++i) w += sel[i * kBulBlock];
    out[j] = has_nan ? kNaN : (w / (float)beta);
  }
}

template <int N, int F>
static int launch_bulyan_fast(const float* const* rows_host, const int32_t* order, int64_t d_all,
                              float* out_all, hipStream_t s) {
  constexpr int MMAX = N - F - 2;
  constexpr int kMaxVec = (MMAX <= 20) ? 4 : (MMAX <= 44 ? 2 : 1);
  int vec = common_vec_width(reinterpret_cast<const void* const*>(rows_host), N, out_all);
  if (vec > kMaxVec) vec = kMaxVec;
  // pieces of at most 2^29 columns so that byte offsets fit 32 bits inside the kernel
  for (int64_t lo = 0; lo < d_all; lo += kMaxColsPerLaunch) {
    const int64_t d = (d_all - lo < kMaxColsPerLaunch) ? (d_all - lo) : kMaxColsPerLaunch;
    RowTable tab{};
    for (int i = 0; i < N; ++i) tab.p[i] = rows_host[i] + lo;
    float* out = out_all + lo;
    if (vec == 4 && kMaxVec >= 4 && d / 4 > 0) {
      const int64_t nvec = d / 4;
      hipLaunchKernelGGL((bulyan_pass2_kernel<N, F, (kMaxVec >= 4 ? 4 : 1)>),
                         dim3(stream_grid(nvec, kBulBlock, kColMaxBlocks)), dim3(kBulBlock), 0,
                         s, tab, order, nvec, 1, out, tuning().bulyan_short, (int)(d - nvec * 4));
    } else if (vec >= 2 && kMaxVec >= 2 && d / 2 > 0) {
      const int64_t nvec = d / 2;
      hipLaunchKernelGGL((bulyan_pass2_kernel<N, F, (kMaxVec >= 2 ? 2 : 1)>),
                         dim3(stream_grid(nvec, kBulBlock, kColMaxBlocks)), dim3(kBulBlock), 0,
                         s, tab, order, nvec, 1, out, tuning().bulyan_short, (int)(d - nvec * 2));
    } else {
      hipLaunchKernelGGL((bulyan_pass2_kernel<N, F, 1>),
                         dim3(stream_grid(d, kBulBlock, kColMaxBlocks)), dim3(kBulBlock), 0,
                         s, tab, order, d, 1, out, tuning().bulyan_short, 0);
    }
    BM_LAUNCH_CHECK();
  }
  return 0;
}

// ---------------------------------------------------------------------------
// Aksel pass 1: median + per-row squared distance to the median.
// ---------------------------------------------------------------------------
// (Beyond ~28 rows the 2 N row pointers do not fit the scalar registers and the compiler parks them in the lanes of vector
// registers — 158 such slots at n = 51 — re-reading them with v_readlane inside the loop; short of vector registers on
// top, the kernel loads 4 bytes per lane there.  Measured alternative, round 6, removed: the table kept by hand in the
// lanes of ONE register pair, two v_readlane per load at a static lane, no spilled SGPR, 4 or 8 bytes per lane: 1 157 /
// 921 us against 638 us for this form at n = 51, d = 11.2 M (profiles/r06_n51_aksel_forms.txt) — every load then waits
// for its own pair of v_readlane results instead of a batch the compiler schedules ahead.)
template <int N, int VEC>
__global__ __launch_bounds__(kColBlock) void aksel_pass1_kernel(RowTable rows, int64_t nvec,
                                                                float* __restrict__ median_out,
                                                                double* __restrict__ partial) {
  __shared__ double red[kColBlock / 64];
  float acc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) acc[i] = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * kColBlock;
  for (int64_t v = (int64_t)blockIdx.x * kColBlock + threadIdx.x; v < nvec; v += stride) {
    float x[VEC][N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float t[VEC];
      load_stream<VEC>(rows.p[i] + v * VEC, t);
#pragma unroll
      for (int c = 0; c < VEC; ++c) x[c][i] = t[c];
    }
    float med[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      float srt[N];
#pragma unroll
      for (int i = 0; i < N; ++i) srt[i] = x[c][i];
      med[c] = column_rule<N, BM_OP_MEDIAN>(srt, 0, 1.0f, nullptr);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float df = x[c][i] - med[c];
        acc[i] += df * df;  // (x - m).pow_(2).sum()
      }
    }
    if (median_out != nullptr) store_stream<VEC>(median_out + v * VEC, med);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double r = block_reduce_sum<kColBlock>((double)acc[i], red);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * BM_MAX_ROWS + i] = r;
  }
}

// sq_out[i] = sum over the workgroups' partials, fixed order: wave w adds the partials of workgroups w, w + 16, ...
// (independent loads, in flight together), then wave 0 adds the 16 wave sums in order.  (Round 2 walked the up to
// 1 024 partials with one dependent load after the other: 232 us, more than the pass it finishes.)
constexpr int kAkselFinWaves = 16;
__global__ __launch_bounds__(64 * kAkselFinWaves) void aksel_finish_kernel(const double* __restrict__ partial,
                                                                          int nparts, int n,
                                                                          double* __restrict__ sq_out) {
  __shared__ double wsum[kAkselFinWaves][BM_MAX_ROWS];
  const int i = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0;
  if (i < n) {
#pragma unroll 8
    for (int b = wave; b < nparts; b += kAkselFinWaves) s += partial[(int64_t)b * BM_MAX_ROWS + i];
  }
  wsum[wave][i] = s;
  __syncthreads();
  if (wave != 0 || i >= n) return;
  double tot = wsum[0][i];
#pragma unroll
  for (int w = 1; w < kAkselFinWaves; ++w) tot += wsum[w][i];
  sq_out[i] = tot;
}

constexpr int kAkselMaxBlocks = 1024;

template <int N>
static int launch_aksel_n(const float* const* rows_host, int64_t d, float* median_out,
                          double* sq_out, double* partial, hipStream_t s) {
  RowTable tab{};
  for (int i = 0; i < N; ++i) tab.p[i] = rows_host[i];
  int vec = common_vec_width(reinterpret_cast<const void* const*>(rows_host), N, median_out);
  constexpr int kMaxVec = (N <= 52) ? 2 : 1;
  if (vec > kMaxVec) vec = kMaxVec;
  if (vec == 2 && N > 28 && tuning().col_wide == 0) vec = 1;  // (BM_COL_WIDE: 8 against 4 bytes per lane beyond 28 rows, A/B)
  int nparts = 0;
  int64_t body = 0;
  if (vec == 2 && kMaxVec >= 2 && d / 2 > 0) {
    const int64_t nvec = d / 2;
    const int grid = stream_grid(nvec, kColBlock, kAkselMaxBlocks - 1);
    hipLaunchKernelGGL((aksel_pass1_kernel<N, (kMaxVec >= 2 ? 2 : 1)>), dim3(grid), dim3(kColBlock), 0, s,
                       tab, nvec, median_out, partial);
    BM_LAUNCH_CHECK();
    nparts = grid;
    body = nvec * 2;
  }
  if (body < d) {
    RowTable tail{};
    for (int i = 0; i < N; ++i) tail.p[i] = rows_host[i] + body;
    const int64_t rest = d - body;
    const int grid = (body == 0) ? stream_grid(rest, kColBlock, kAkselMaxBlocks) : 1;
    hipLaunchKernelGGL((aksel_pass1_kernel<N, 1>), dim3(grid), dim3(kColBlock), 0, s, tail, rest,
                       median_out ? median_out + body : nullptr,
                       partial + (int64_t)nparts * BM_MAX_ROWS);
    BM_LAUNCH_CHECK();
    nparts += grid;
  }
  hipLaunchKernelGGL(aksel_finish_kernel, dim3(1), dim3(64 * kAkselFinWaves), 0, s, partial, nparts, N, sq_out);
  BM_LAUNCH_CHECK();
  return 0;
}

template <int... Ns>
static int dispatch_aksel(std::integer_sequence<int, Ns...>, const float* const* rows, int n,
                          int64_t d, float* median_out, double* sq_out, double* partial,
                          hipStream_t s) {
  int rc = BM_EINVAL;
  ((n == Ns + 1 ? (rc = launch_aksel_n<Ns + 1>(rows, d, median_out, sq_out, partial, s), 0) : 0), ...);
  return rc;
}

int64_t pairwise_workspace_bytes(int n, int64_t d);  // pairwise.hip

}  // namespace bm

extern "C" int bm_bulyan_pass2(const float* const* rows, int n, const int32_t* order, int f, int m,
                               int64_t d, float* out, void* stream) {
  using namespace bm;
  if (rows == nullptr || order == nullptr || (out == nullptr && d > 0) || n < 1 || n > BM_MAX_ROWS || f < 1 ||
      n < 4 * f + 3 || m < 1 || m > n - f - 2 || d < 0)
    return BM_EINVAL;
  if (d == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int m_max = n - f - 2;
  if (m == m_max) {
    // register-resident instances for the (n, f) grid the reference exercises
    // (reproduce.py:139,182; reproduce-appendix.py:121-158) and their neighbours
#define BM_BULYAN_CASE(NN, FF) \
  if (n == NN && f == FF) return launch_bulyan_fast<NN, FF>(rows, order, d, out, s);
    BM_BULYAN_CASE(11, 2)
    BM_BULYAN_CASE(15, 3)
    BM_BULYAN_CASE(19, 4)
    BM_BULYAN_CASE(25, 5)
    BM_BULYAN_CASE(23, 5)
    BM_BULYAN_CASE(27, 6)
    BM_BULYAN_CASE(31, 7)
    BM_BULYAN_CASE(35, 8)
    BM_BULYAN_CASE(39, 9)
    BM_BULYAN_CASE(43, 10)
    BM_BULYAN_CASE(47, 11)
    BM_BULYAN_CASE(51, 12)
    BM_BULYAN_CASE(51, 10)
    BM_BULYAN_CASE(7, 1)
#undef BM_BULYAN_CASE
  }
  RowTable tab{};
  for (int i = 0; i < n; ++i) tab.p[i] = rows[i];
  const int theta = n - 2 * f - 2;
  const size_t lds = (size_t)(m_max + theta) * kBulBlock * sizeof(float);
  const int grid = stream_grid(d, kBulBlock, kColMaxBlocks);
  if (const int rc = lds_opt_in(reinterpret_cast<const void*>(bulyan_pass2_generic_kernel), lds, BM_MAX_ROWS * sizeof(float*)))
    return rc;
  hipLaunchKernelGGL(bulyan_pass2_generic_kernel, dim3(grid), dim3(kBulBlock), lds, s, tab, order, n,
                     f, m, d, out);
  BM_LAUNCH_CHECK();
  return 0;
}

// (the worker counts of reproduce.py:122-209 with their largest f, like bm_colwise_eval)
extern "C" int bm_bulyan_pass2_eval_supported(int n, int f, int m) {
  return ((n == 11 && f == 2) || (n == 25 && f == 5) || (n == 51 && f == 12)) && m == n - f - 2 ? 1 : 0;
}

extern "C" int bm_bulyan_pass2_eval(const float* const* honests, int h, int copies, const int32_t* order, int f, int m,
                                    int64_t d, const float* avg, const float* dir, float t, const double* t_dev, double* out,
                                    void* ws, void* stream) {
  using namespace bm;
  const int n = h + copies;
  if (honests == nullptr || order == nullptr || out == nullptr || ws == nullptr || h < 1 || copies < 1 || d < 0 ||
      d > kMaxColsPerLaunch || (d > 0 && (avg == nullptr || dir == nullptr)) || !bm_bulyan_pass2_eval_supported(n, f, m))
    return BM_EINVAL;
  for (int i = 0; i < h && d > 0; ++i)
    if (honests[i] == nullptr) return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* partial = static_cast<double*>(ws);
  if (n == 11) return launch_bulyan_eval<11, 2>(honests, h, order, d, avg, dir, t, t_dev, out, partial, s);
  if (n == 25) return launch_bulyan_eval<25, 5>(honests, h, order, d, avg, dir, t, t_dev, out, partial, s);
  return launch_bulyan_eval<51, 12>(honests, h, order, d, avg, dir, t, t_dev, out, partial, s);
}

extern "C" int bm_aksel_pass1(const float* const* rows, int n, int64_t d, float* median_out,
                              double* sq_out, void* ws, void* stream) {
  using namespace bm;
  // d == 0 (an empty shard) is legal: no partials, the finish kernel writes zeros
  if (rows == nullptr || sq_out == nullptr || ws == nullptr || n < 1 || n > BM_MAX_ROWS || d < 0)
    return BM_EINVAL;
  return dispatch_aksel(std::make_integer_sequence<int, BM_MAX_ROWS>{}, rows, n, d, median_out,
                        sq_out, static_cast<double*>(ws), static_cast<hipStream_t>(stream));
}

namespace bm {
int64_t study_workspace_bytes();  // study.hip
}

extern "C" int64_t bm_workspace_bytes(int kind, int n, int64_t d) {
  using namespace bm;
  if (n < 1 || n > BM_MAX_ROWS) return BM_EINVAL;
  switch (kind) {
    case BM_WS_PAIRWISE:
      return pairwise_workspace_bytes(n, d);
    case BM_WS_AKSEL:
      return (int64_t)kAkselMaxBlocks * BM_MAX_ROWS * (int64_t)sizeof(double);
    case BM_WS_STATS:
      return (int64_t)2048 * 3 * (int64_t)sizeof(double);
    case BM_WS_DOT:
      return (int64_t)1025 * 42 * (int64_t)sizeof(double);
    case BM_WS_STEP:
      return (int64_t)16385 * 6 * (int64_t)sizeof(double);  // kStepMaxBlocks + 1 sets of 6 partials (step.hip)
    case BM_WS_STUDY:
      return study_workspace_bytes();
    default:
      return BM_EINVAL;
  }
}
