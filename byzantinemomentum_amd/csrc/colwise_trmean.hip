// colwise_trmean.hip — the instances of one coordinate-wise rule (contract: colwise.hip, launch logic: colwise_dispatch.h).
#include "colwise_dispatch.h"

namespace bm {
int colwise_trmean(const float* const* rows, int n, int64_t d, int f, float* out, hipStream_t stream) {
  return colwise_dispatch<BM_OP_TRMEAN>(rows, n, d, f, out, stream);
}
}  // namespace bm
