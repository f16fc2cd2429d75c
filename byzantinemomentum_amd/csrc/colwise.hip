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

#include "bm_common.h"

namespace bm {
// one translation unit per rule (colwise_<rule>.hip, launch logic in colwise_dispatch.h)
int colwise_median(const float* const* rows, int n, int64_t d, int f, float* out, hipStream_t stream);
int colwise_trmean(const float* const* rows, int n, int64_t d, int f, float* out, hipStream_t stream);
int colwise_phocas(const float* const* rows, int n, int64_t d, int f, float* out, hipStream_t stream);
int colwise_meamed(const float* const* rows, int n, int64_t d, int f, float* out, hipStream_t stream);
}  // namespace bm

extern "C" int bm_colwise(int op, const float* const* rows, int n, int64_t d, int f, float* out,
                          void* stream) {
  using namespace bm;
  if (rows == nullptr || (out == nullptr && d > 0) || n < 1 || n > BM_MAX_ROWS || d < 0) return BM_EINVAL;
  if (d == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (op) {
    case BM_OP_MEDIAN:
      return colwise_median(rows, n, d, 0, out, s);
    case BM_OP_TRMEAN:
      if (f < 0 || n < 2 * f + 1) return BM_EINVAL;
      return colwise_trmean(rows, n, d, f, out, s);
    case BM_OP_PHOCAS:
      if (f < 0 || n < 2 * f + 1) return BM_EINVAL;
      return colwise_phocas(rows, n, d, f, out, s);
    case BM_OP_MEAMED:
      if (f < 0 || n < 2 * f + 1) return BM_EINVAL;
      return colwise_meamed(rows, n, d, f, out, s);
    default:
      return BM_EINVAL;
  }
}
