// colwise.hip — coordinate-wise rules over the worker axis, one HBM pass.
//
// Replaces (reference, PyTorch):
//   median   aggregators/median.py:39    torch.stack(g).median(dim=0)[0]
//   trmean   aggregators/trmean.py:33    g.sort(dim=0).values[f:-f].mean(dim=0)
//   phocas   aggregators/trmean.py:81-94  closest(g, f, trmean(g, f))
//   meamed   aggregators/trmean.py:96-109 closest(g, f, median(g))
//
// Layout: no stacked copy exists.  The n rows are read through a by-value pointer
// table; a lane owns VEC consecutive coordinates, so a wave issues n independent
// 64*VEC*4-byte fully coalesced non-temporal loads per step, keeps the n values of
// each of its columns in VGPRs, sorts them with a compile-time merge-exchange
// network (v_min_f32/v_max_f32, no LDS, no divergence) and applies the rule's
// reduction before one coalesced store.  Algorithmic traffic: 4*d*(n+1) bytes.
//
// NaN semantics follow the torch in this image (2.10): `median` propagates NaN,
// `sort` orders NaN last (so trmean is NaN iff more than f values of the column are NaN).

#include "colwise_kernels.h"

namespace bm {

constexpr int kBurstMaxRows = 25;  // 4 waves per SIMD (1024 lanes per CU) leave 128 VGPRs: trmean at n = 25 just fits, n = 26 spills

template <int N, int OP, int VEC>
static int launch_colwise_vec(const RowTable& rows_all, int64_t d_all, int f, float* out_all,
                              hipStream_t stream) {
  const int keep = (OP == BM_OP_TRMEAN) ? (N - 2 * f) : (N - f);
  const float inv_keep = 1.0f / (float)(keep > 0 ? keep : 1);
  // pieces of at most 2^29 columns so that byte offsets fit 32 bits inside the kernel
  for (int64_t lo = 0; lo < d_all; lo += kMaxColsPerLaunch) {
    const int64_t d = (d_all - lo < kMaxColsPerLaunch) ? (d_all - lo) : kMaxColsPerLaunch;
    RowTable rows = rows_all;
    for (int i = 0; i < N; ++i) rows.p[i] += lo;
    const int64_t nvec = d / VEC;
    const int tail = (int)(d - nvec * VEC);
    if constexpr (VEC == 4 && N <= kBurstMaxRows && (OP == BM_OP_MEDIAN || OP == BM_OP_TRMEAN)) {
      // burst form: one workgroup per CU; worth it once every CU has several iterations to stage
      const int cus = compute_units();
      const int64_t burst_iters = nvec / ((int64_t)cus * kBurstThreads);
      if (tuning().col_burst > 0 && burst_iters >= tuning().col_burst) {
        hipLaunchKernelGGL((colwise_burst_kernel<N, OP, VEC>), dim3(cus), dim3(kBurstThreads), 0, stream, rows, nvec,
                           tail, f, inv_keep, out_all + lo);
        BM_LAUNCH_CHECK();
        continue;
      }
    }
    const int grid = stream_grid(nvec, kColBlock, kColMaxBlocks);
    hipLaunchKernelGGL((colwise_kernel<N, OP, VEC>), dim3(grid), dim3(kColBlock), 0, stream, rows,
                       nvec, tail, f, inv_keep, 1, out_all + lo);
    BM_LAUNCH_CHECK();
  }
  return 0;
}

// One launch: vector body with the widest vector the pointers allow, the d % VEC trailing
// columns are handled by the last workgroup of the same kernel.
template <int N, int OP>
static int launch_colwise_n(const float* const* rows_host, int64_t d, int f, float* out,
                            hipStream_t stream) {
  RowTable tab{};
  for (int i = 0; i < N; ++i) tab.p[i] = rows_host[i];
  // Register budget: N*VEC live values.  Keep it at or below ~112 so that >= 4 waves/SIMD fit.
  int vec = common_vec_width(reinterpret_cast<const void* const*>(rows_host), N, out);
  constexpr int kMaxVec = (N <= 28) ? 4 : (N <= 56 ? 2 : 1);
  if (vec > kMaxVec) vec = kMaxVec;
  if (vec == 4 && kMaxVec >= 4)
    return launch_colwise_vec < N, OP, (kMaxVec >= 4 ? 4 : 1) > (tab, d, f, out, stream);
  if (vec == 2 && kMaxVec >= 2)
    return launch_colwise_vec < N, OP, (kMaxVec >= 2 ? 2 : 1) > (tab, d, f, out, stream);
  return launch_colwise_vec<N, OP, 1>(tab, d, f, out, stream);
}

template <int OP, int... Ns>
static int dispatch_n(std::integer_sequence<int, Ns...>, const float* const* rows, int n,
                      int64_t d, int f, float* out, hipStream_t stream) {
  int rc = BM_EINVAL;
  // Ns = 0..63 -> N = Ns+1
  ((n == Ns + 1 ? (rc = launch_colwise_n<Ns + 1, OP>(rows, d, f, out, stream), 0) : 0), ...);
  return rc;
}

}  // namespace bm

extern "C" int bm_colwise(int op, const float* const* rows, int n, int64_t d, int f, float* out,
                          void* stream) {
  using namespace bm;
  if (rows == nullptr || (out == nullptr && d > 0) || n < 1 || n > BM_MAX_ROWS || d < 0) return BM_EINVAL;
  if (d == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto seq = std::make_integer_sequence<int, BM_MAX_ROWS>{};
  switch (op) {
    case BM_OP_MEDIAN:
      return dispatch_n<BM_OP_MEDIAN>(seq, rows, n, d, 0, out, s);
    case BM_OP_TRMEAN:
      if (f < 0 || n < 2 * f + 1) return BM_EINVAL;
      return dispatch_n<BM_OP_TRMEAN>(seq, rows, n, d, f, out, s);
    case BM_OP_PHOCAS:
      if (f < 0 || n < 2 * f + 1) return BM_EINVAL;
      return dispatch_n<BM_OP_PHOCAS>(seq, rows, n, d, f, out, s);
    case BM_OP_MEAMED:
      if (f < 0 || n < 2 * f + 1) return BM_EINVAL;
      return dispatch_n<BM_OP_MEAMED>(seq, rows, n, d, f, out, s);
    default:
      return BM_EINVAL;
  }
}
