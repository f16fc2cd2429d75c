// gram_tail.h — what the last workgroup of a folded Gram launch does (gram_bf16.hip; filled by pairwise.hip).
#pragma once
#include <stdint.h>

namespace bm {

struct GramTail {
  double* gpart;      // NULL: no tail (the reduction is a launch of its own, gram_reduce_sqdist_kernel)
  double* gram;
  double* sq;
  double tau;
  int n_full;
  int rank;           // 1: rank when the gate lists nothing
  int rank_f, rank_m, rank_mode;
  int32_t* order;
  double* scores;
};

}  // namespace bm
