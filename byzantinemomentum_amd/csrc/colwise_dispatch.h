// colwise_dispatch.h — launch logic of the coordinate-wise rules (see colwise.hip for the contract), shared by the
// four translation units colwise_median / _trmean / _phocas / _meamed.hip: one rule each, n = 1..64 rows, so that the
// 768 kernel instances compile side by side (one file took 136 s of a 140 s build).
#pragma once
#include "colwise_kernels.h"

namespace bm {

constexpr int kBurstMaxRows = 25;  // 4 waves per SIMD (1024 lanes per CU) leave 128 VGPRs: trmean at n = 25 just fits, n = 26 spills
constexpr int kBurstMaxRowsClosest = 22;  // phocas / meamed keep the centre and the window search live on top: n = 23 spills

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
    // (every rule: none needs the LDS for itself any more)
    if constexpr (VEC == 4 && N <= ((OP == BM_OP_MEDIAN || OP == BM_OP_TRMEAN) ? kBurstMaxRows : kBurstMaxRowsClosest)) {
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
  // (phocas / meamed at n = 55, 56 with two columns per lane, and beyond 28 rows with four: the unrolled window instances
  //  exceed the unroller's budget and the column arrays land in scratch)
  constexpr bool kClosest = (OP == BM_OP_PHOCAS || OP == BM_OP_MEAMED);
  constexpr int kMaxVec = (N <= 28) ? 4 : ((!kClosest && N <= 52) ? 4 : (N <= (kClosest ? 54 : 56) ? 2 : 1));
  if (vec > kMaxVec) vec = kMaxVec;
  // 29-52 rows, median / trimmed mean: 16-byte columns (2 waves per SIMD at ~210-240 VGPRs) against 8-byte ones (3 waves
  // at ~165): BM_COL_WIDE (default 1) — measured at n = 51, profiles/r06_n51_wide_columns.txt
  if (vec == 4 && N > 28 && tuning().col_wide == 0) vec = 2;
  if (vec == 4 && kMaxVec >= 4)
    return launch_colwise_vec < N, OP, (kMaxVec >= 4 ? 4 : 1) > (tab, d, f, out, stream);
  if (vec >= 2 && kMaxVec >= 2)
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


// the rule OP for any n = 1..64
template <int OP>
static int colwise_dispatch(const float* const* rows, int n, int64_t d, int f, float* out, hipStream_t stream) {
  return dispatch_n<OP>(std::make_integer_sequence<int, BM_MAX_ROWS>{}, rows, n, d, f, out, stream);
}

}  // namespace bm
