// search_eval.hip — ONE candidate of the attacks' factor search against a coordinate-wise rule, evaluate only.
//
// attacks/identical.py:67-77 maximises, over the factor t,
//     eval(t) = | GAR(honests + [avg + t * dir] * k) - avg |^2          (identical.py:73-76)
// with tools.line_maximize (tools/misc.py:468-514): by default 16 evaluations per step, each one a full aggregation.
// Krum, Brute, the average, the median and Bulyan have cheaper forms of the whole search (linesearch.cpp, step.py);
// the trimmed mean, phocas and meamed do not — their output is not a function of inner products or of two order
// statistics — and cost, per candidate, the candidate vector (2 rows read, 1 written), the rule (h + 1 rows read, the
// k aliased copies from cache, 1 written) and the objective (2 rows read): h + 5 rows read, 2 written, three launches.
//
// Here the candidate exists in registers only: a lane reads its columns of the h honest rows, of avg and of dir,
// forms avg + t * dir with the arithmetic of bm_multi_fma3 (fma(t, dir, 1 * avg): the same bits as the vector the
// generic form writes), runs the rule's own device code (column_rule, colwise_kernels.h) on the h + k values, and
// accumulates (rule - avg)^2 — h + 2 rows read, NOTHING written, one launch.  The value of the rule at every column is
// the one bm_colwise gives on the materialised stack, so the objective agrees with the generic form to the rounding
// of its sum (fp32 over <= 64 elements per lane, fp64 beyond; the generic form takes it from the distance kernel).
//
// Instances: n = h + k in {11, 25, 51} (the worker counts of reproduce.py:122-209, reproduce-appendix.py:109), any
// split into honest rows and copies; other shapes take the generic form (bm_colwise_eval_supported says which).
//
// Also here, for the other searches (ABI 22): the median's own — its candidates are the middle of (lo, hi, candidate),
// lo / hi two order statistics of the honest rows formed once per search by order_pair_kernel and every candidate one
// launch of colwise_eval_kernel<3, MEDIAN> — and sqdist2_kernel, the objective |rule - avg|^2 of a candidate whose rule
// ran in full (Aksel, CGE, Brute, any rule through the generic form).  Bulyan's evaluate-only second pass lives next to
// the pass it mirrors (bulyan.hip, bm_bulyan_pass2_eval).
#include "colwise_kernels.h"

namespace bm {

template <int N, int OP, int VEC>
__global__ __launch_bounds__(kColBlock) void colwise_eval_kernel(RowTable rows, int h, const float* __restrict__ avg,
                                                                 const float* __restrict__ dir, float t_host,
                                                                 const double* __restrict__ t_dev, int64_t nvec,
                                                                 int f, float inv_keep, double* __restrict__ partial) {
  // the candidate's factor: the caller's number, or the one the device cursor left in device memory (bm_colwise_eval_tdev;
  // rounded to fp32 as the host's conversion of the same number would be)
  const float t = t_dev != nullptr ? (float)t_dev[0] : t_host;
  __shared__ double red[kColBlock / 64];
  float* const lds = nullptr;  // (column_rule needs no LDS)
  float acc = 0.0f;
  double wide = 0.0;
  int since = 0;
  const int64_t stride = (int64_t)gridDim.x * kColBlock;
  for (int64_t v = (int64_t)blockIdx.x * kColBlock + threadIdx.x; v < nvec; v += stride) {
    const int64_t j = v * VEC;
    float x[VEC][N];
    float a[VEC], dr[VEC];
    load_stream<VEC>(avg + j, a);
    load_stream<VEC>(dir + j, dr);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i < h) {  // wave-uniform
        float tmp[VEC];
        load_stream<VEC>(rows.p[i] + j, tmp);
#pragma unroll
        for (int c = 0; c < VEC; ++c) x[c][i] = tmp[c];
      }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      const float cand = __builtin_fmaf(t, dr[c], 1.0f * a[c]);  // bm_multi_fma3(out, avg, dir, 1, t): same bits
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i >= h) x[c][i] = cand;
      const float r = column_rule<N, OP>(x[c], f, inv_keep, lds);
      const float df = r - a[c];  // aggregated.sub_(grad_avg) (identical.py:75)
      acc = __builtin_fmaf(df, df, acc);
    }
    if (++since == 16) {
      wide += (double)acc;
      acc = 0.0f;
      since = 0;
    }
  }
  const double r = block_reduce_sum<kColBlock>(wide + (double)acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <int N, int OP>
static int launch_eval(const float* const* rows, int h, int64_t d, int f, const float* avg, const float* dir, float t,
                       const double* t_dev, double* out, double* partial, hipStream_t s) {
  constexpr int kMaxVec = (N <= 28) ? 4 : 2;
  const int keep = (OP == BM_OP_TRMEAN) ? (N - 2 * f) : (N - f);
  const float inv_keep = 1.0f / (float)(keep > 0 ? keep : 1);
  RowTable tab{};
  for (int i = 0; i < h; ++i) tab.p[i] = rows[i];
  const void* more[2] = {avg, dir};
  int vec = common_vec_width(reinterpret_cast<const void* const*>(rows), h, nullptr);
  const int vec2 = common_vec_width(more, 2, nullptr);
  if (vec2 < vec) vec = vec2;
  if (vec > kMaxVec) vec = kMaxVec;
  int nparts = 0;
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    const int grid = stream_grid(nvec, kColBlock, kEvalMaxBlocks);
    auto kern = vec == 4 ? colwise_eval_kernel<N, OP, (kMaxVec >= 4 ? 4 : 2)> : colwise_eval_kernel<N, OP, 2>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kColBlock), 0, s, tab, h, avg, dir, t, t_dev, nvec, f, inv_keep, partial);
    BM_LAUNCH_CHECK();
    nparts = grid;
    body = nvec * vec;
  }
  if (body < d) {
    RowTable tail{};
    for (int i = 0; i < h; ++i) tail.p[i] = rows[i] + body;
    const int64_t rest = d - body;
    const int grid = (body == 0) ? stream_grid(rest, kColBlock, kEvalMaxBlocks) : 1;
    hipLaunchKernelGGL((colwise_eval_kernel<N, OP, 1>), dim3(grid), dim3(kColBlock), 0, s, tail, h, avg + body, dir + body,
                       t, t_dev, rest, f, inv_keep, partial + nparts);
    BM_LAUNCH_CHECK();
    nparts += grid;
  }
  // d == 0: no partial, the finish kernel writes zero (every rank of a sharded job reaches its all-reduce)
  hipLaunchKernelGGL(eval_finish_kernel<kEvalFinishThreads>, dim3(1), dim3(kEvalFinishThreads), 0, s, partial, nparts, out);
  BM_LAUNCH_CHECK();
  return 0;
}

template <int N>
static int launch_eval_op(int op, const float* const* rows, int h, int64_t d, int f, const float* avg, const float* dir,
                          float t, const double* t_dev, double* out, double* partial, hipStream_t s) {
  switch (op) {
    case BM_OP_TRMEAN: return launch_eval<N, BM_OP_TRMEAN>(rows, h, d, f, avg, dir, t, t_dev, out, partial, s);
    case BM_OP_PHOCAS: return launch_eval<N, BM_OP_PHOCAS>(rows, h, d, f, avg, dir, t, t_dev, out, partial, s);
    case BM_OP_MEAMED: return launch_eval<N, BM_OP_MEAMED>(rows, h, d, f, avg, dir, t, t_dev, out, partial, s);
    default: return BM_EINVAL;
  }
}

// |a - b|^2 of two vectors with the accumulation of colwise_eval_kernel (fp32 over 16 elements per lane, fp64 beyond, the
// partials added in a fixed order): the objective of a candidate whose rule has no evaluate-only form
// (aggregated.sub_(grad_avg); aggregated.dot(aggregated), identical.py:75-76).  The n x n machinery of the distance pass
// costs 64 us for these two rows at d = 11.2 M (Gram kernel 45 + reduction + gated launch); this is one 15 us pass.
template <int VEC>
__global__ __launch_bounds__(kColBlock) void sqdist2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            int64_t nvec, double* __restrict__ partial) {
  __shared__ double red[kColBlock / 64];
  float acc = 0.0f;
  double wide = 0.0;
  int since = 0;
  const int64_t stride = (int64_t)gridDim.x * kColBlock;
  for (int64_t v = (int64_t)blockIdx.x * kColBlock + threadIdx.x; v < nvec; v += stride) {
    float x[VEC], y[VEC];
    load_stream<VEC>(a + v * VEC, x);
    load_stream<VEC>(b + v * VEC, y);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      const float df = x[c] - y[c];
      acc = __builtin_fmaf(df, df, acc);
    }
    if (++since == 16) {
      wide += (double)acc;
      acc = 0.0f;
      since = 0;
    }
  }
  const double r = block_reduce_sum<kColBlock>(wide + (double)acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// Two order statistics of the h honest values of every column in ONE pass over the rows: the `lo` / `hi` of the median's
// own factor search (step.py).  median(honests + [b] * k) = middle of (b, lo, hi) with lo / hi the medians of the honest
// values and k copies of -inf / +inf (median.py:31-39 on either stack): in sorted order the -inf copies come first and
// the +inf copies last, so lo is the honest value of rank (n-1)/2 - k (-inf below rank 0) and hi the one of rank
// (n-1)/2 (+inf beyond rank h-1) — values of the rows, no arithmetic, hence the bits of the two median calls they
// replace (2 (n + 1) row passes, the 2 k aliased copies among them, against h + 2 here).  A NaN anywhere in the column
// makes both NaN, as torch.median does for either stack.  The rows are padded with +inf to the network's size N; the
// rank is wave-uniform and picked by a chain of selects at static register indices.
template <int N, int VEC>
__global__ __launch_bounds__(kColBlock) void order_pair_kernel(RowTable rows, int h, int il, int ih, int64_t nvec,
                                                               float* __restrict__ lo, float* __restrict__ hi) {
  const float kInf = __builtin_inff();
  const float kNaN = __builtin_nanf("");
  const int64_t stride = (int64_t)gridDim.x * kColBlock;
  for (int64_t v = (int64_t)blockIdx.x * kColBlock + threadIdx.x; v < nvec; v += stride) {
    const int64_t j = v * VEC;
    float x[VEC][N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i < h) {  // wave-uniform
        float tmp[VEC];
        load_stream<VEC>(rows.p[i] + j, tmp);
#pragma unroll
        for (int c = 0; c < VEC; ++c) x[c][i] = tmp[c];
      } else {
#pragma unroll
        for (int c = 0; c < VEC; ++c) x[c][i] = kInf;
      }
    }
    float a[VEC], b[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      bool has_nan = false;
#pragma unroll
      for (int i = 0; i < N; ++i) has_nan |= (x[c][i] != x[c][i]);
      sort_network<N>(x[c]);
      float va = -kInf, vb = kInf;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        va = (i == il) ? x[c][i] : va;
        vb = (i == ih) ? x[c][i] : vb;
      }
      a[c] = has_nan ? kNaN : va;
      b[c] = has_nan ? kNaN : vb;
    }
    store_stream<VEC>(lo + j, a);
    store_stream<VEC>(hi + j, b);
  }
}

template <int N>
static int launch_order_pair(const float* const* rows, int h, int64_t d, int il, int ih, float* lo, float* hi,
                             hipStream_t s) {
  constexpr int kMaxVec = (N <= 28) ? 4 : 2;
  RowTable tab{};
  for (int i = 0; i < h; ++i) tab.p[i] = rows[i];
  const void* outs[2] = {lo, hi};
  int vec = common_vec_width(reinterpret_cast<const void* const*>(rows), h, nullptr);
  const int vec2 = common_vec_width(outs, 2, nullptr);
  if (vec2 < vec) vec = vec2;
  if (vec > kMaxVec) vec = kMaxVec;
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    const int grid = stream_grid(nvec, kColBlock, kColMaxBlocks);
    auto kern = vec == 4 ? order_pair_kernel<N, (kMaxVec >= 4 ? 4 : 2)> : order_pair_kernel<N, 2>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kColBlock), 0, s, tab, h, il, ih, nvec, lo, hi);
    BM_LAUNCH_CHECK();
    body = nvec * vec;
  }
  if (body < d) {
    RowTable tail{};
    for (int i = 0; i < h; ++i) tail.p[i] = rows[i] + body;
    const int64_t rest = d - body;
    const int grid = (body == 0) ? stream_grid(rest, kColBlock, kColMaxBlocks) : 1;
    hipLaunchKernelGGL((order_pair_kernel<N, 1>), dim3(grid), dim3(kColBlock), 0, s, tail, h, il, ih, rest, lo + body,
                       hi + body);
    BM_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace bm

extern "C" int bm_order_pair_supported(int h) { return h >= 1 && h <= 51 ? 1 : 0; }

extern "C" int bm_order_pair(const float* const* rows, int h, int64_t d, int il, int ih, float* lo, float* hi,
                             void* stream) {
  using namespace bm;
  if (rows == nullptr || !bm_order_pair_supported(h) || d < 0) return BM_EINVAL;
  if (d == 0) return 0;
  if (lo == nullptr || hi == nullptr || lo == hi) return BM_EINVAL;
  for (int i = 0; i < h; ++i)
    if (rows[i] == nullptr) return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (h <= 11) return launch_order_pair<11>(rows, h, d, il, ih, lo, hi, s);
  if (h <= 25) return launch_order_pair<25>(rows, h, d, il, ih, lo, hi, s);
  return launch_order_pair<51>(rows, h, d, il, ih, lo, hi, s);
}

// (BM_OP_MEDIAN, n = 3: the median's own search — every candidate is the middle of (candidate, lo, hi), lo / hi two order
//  statistics of the honest rows formed once per search, step.py — evaluated without writing candidate or median)
extern "C" int bm_colwise_eval_supported(int op, int n) {
  if (op == BM_OP_MEDIAN) return n == 3 ? 1 : 0;
  return (op == BM_OP_TRMEAN || op == BM_OP_PHOCAS || op == BM_OP_MEAMED) && (n == 11 || n == 25 || n == 51) ? 1 : 0;
}

extern "C" int bm_sqdist2(const float* a, const float* b, int64_t d, double* out, void* ws, void* stream) {
  using namespace bm;
  if (out == nullptr || ws == nullptr || d < 0 || (d > 0 && (a == nullptr || b == nullptr))) return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* partial = static_cast<double*>(ws);
  const void* both[2] = {a, b};
  const int vec = d > 0 ? common_vec_width(both, 2, nullptr) : 1;
  int nparts = 0;
  int64_t body = 0;
  if (vec >= 2 && d / vec > 0) {
    const int64_t nvec = d / vec;
    const int grid = stream_grid(nvec, kColBlock, kEvalMaxBlocks);
    if (vec == 4)
      hipLaunchKernelGGL(sqdist2_kernel<4>, dim3(grid), dim3(kColBlock), 0, s, a, b, nvec, partial);
    else
      hipLaunchKernelGGL(sqdist2_kernel<2>, dim3(grid), dim3(kColBlock), 0, s, a, b, nvec, partial);
    BM_LAUNCH_CHECK();
    nparts = grid;
    body = nvec * vec;
  }
  if (body < d) {
    const int64_t rest = d - body;
    const int grid = (body == 0) ? stream_grid(rest, kColBlock, kEvalMaxBlocks) : 1;
    hipLaunchKernelGGL(sqdist2_kernel<1>, dim3(grid), dim3(kColBlock), 0, s, a + body, b + body, rest, partial + nparts);
    BM_LAUNCH_CHECK();
    nparts += grid;
  }
  // d == 0: no partial, the finish kernel writes zero (every rank of a sharded job reaches its all-reduce)
  hipLaunchKernelGGL(eval_finish_kernel<kEvalFinishThreads>, dim3(1), dim3(kEvalFinishThreads), 0, s, partial, nparts, out);
  BM_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t bm_colwise_eval_workspace_bytes(void) { return (int64_t)(2 * bm::kEvalMaxBlocks) * (int64_t)sizeof(double); }

static int colwise_eval_call(int op, const float* const* honests, int h, int copies, int64_t d, int f, const float* avg,
                             const float* dir, float t, const double* t_dev, double* out, void* ws, void* stream) {
  using namespace bm;
  const int n = h + copies;
  if (honests == nullptr || out == nullptr || ws == nullptr || h < 1 || copies < 1 || n > BM_MAX_ROWS || d < 0 || f < 0 ||
      n < 2 * f + 1 || (d > 0 && (avg == nullptr || dir == nullptr)) || !bm_colwise_eval_supported(op, n))
    return BM_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* partial = static_cast<double*>(ws);
  switch (n) {
    case 3: return launch_eval<3, BM_OP_MEDIAN>(honests, h, d, f, avg, dir, t, t_dev, out, partial, s);
    case 11: return launch_eval_op<11>(op, honests, h, d, f, avg, dir, t, t_dev, out, partial, s);
    case 25: return launch_eval_op<25>(op, honests, h, d, f, avg, dir, t, t_dev, out, partial, s);
    default: return launch_eval_op<51>(op, honests, h, d, f, avg, dir, t, t_dev, out, partial, s);
  }
}

extern "C" int bm_colwise_eval(int op, const float* const* honests, int h, int copies, int64_t d, int f,
                               const float* avg, const float* dir, float t, double* out, void* ws, void* stream) {
  return colwise_eval_call(op, honests, h, copies, d, f, avg, dir, t, nullptr, out, ws, stream);
}

extern "C" int bm_colwise_eval_tdev(int op, const float* const* honests, int h, int copies, int64_t d, int f,
                                    const float* avg, const float* dir, const double* t_dev, double* out, void* ws,
                                    void* stream) {
  if (t_dev == nullptr) return BM_EINVAL;
  return colwise_eval_call(op, honests, h, copies, d, f, avg, dir, 0.0f, t_dev, out, ws, stream);
}
