// api.cpp — ABI bookkeeping and host-only entry points of libbm_gar.so.
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "bm_common.h"

namespace bm {
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  if (v == nullptr || *v == '\0') return dflt;
  return atoi(v);
}
static double env_double(const char* name, double dflt) {
  const char* v = getenv(name);
  if (v == nullptr || *v == '\0') return dflt;
  return atof(v);
}
const Tuning& tuning() {
  static const Tuning t = {env_int("BM_COL_BURST", 8),  env_int("BM_MEAN_BURST", 8),
                           env_int("BM_STEP_BURST", 8), env_int("BM_STEP_STREAM", 0), env_int("BM_PAIR_MODE", 0),
                           env_int("BM_PAIR_PLANES", 0), env_int("BM_PAIR_DITHER", 0), env_double("BM_PAIR_TAU", 2e-3)};
  return t;
}
}  // namespace bm

extern "C" int bm_abi_version(void) { return 14; }

extern "C" const char* bm_error_string(int code) {
  if (code == 0) return "success";
  if (code == BM_EINVAL) return "invalid argument for libbm_gar";
  if (code == BM_ENOCOMM) return "RCCL is not available in this process (librccl.so.1 could not be bound)";
  if (code == BM_ECOMM) return "an RCCL call failed";
  if (code < 0) return hipGetErrorString(static_cast<hipError_t>(-code));
  return "unknown libbm_gar status";
}

// Host-side exhaustive subset search of the Brute rule (aggregators/brute.py:47-68).
// Depth-first enumeration in lexicographic order with the running diameter carried down the
// recursion, so a subset is abandoned as soon as its partial diameter can no longer win —
// the visiting ORDER and the strict '<' acceptance are those of the reference, hence the
// same "first smallest" subset is returned.
namespace {
struct BruteSearch {
  const double* dist;
  int n, k;
  std::vector<int> cur, best;
  double best_diam;
  bool have_best;
  void rec(int depth, int start, double diam) {
    if (depth == k) {
      if (!have_best || diam < best_diam) {
        best = cur;
        best_diam = diam;
        have_best = true;
      }
      return;
    }
    // need k-depth more elements from [start, n)
    for (int c = start; c <= n - (k - depth); ++c) {
      double dm = diam;
      bool ok = true;
      for (int t = 0; t < depth; ++t) {
        const double v = dist[cur[t] * n + c];
        if (!isfinite(v)) {
          ok = false;
          break;
        }
        if (v > dm) dm = v;
      }
      if (!ok) continue;
      // pruning: a strictly better subset needs diam < best_diam; dm only grows deeper down
      if (have_best && !(dm < best_diam)) continue;
      cur[depth] = c;
      rec(depth + 1, c + 1, dm);
    }
  }
};
}  // namespace

extern "C" int bm_brute_select(const double* dist_nxn, int n, int f, int32_t* sel_out) {
  if (dist_nxn == nullptr || sel_out == nullptr || n < 1 || n > BM_MAX_ROWS || f < 0 || n - f < 1)
    return BM_EINVAL;
  BruteSearch s;
  s.dist = dist_nxn;
  s.n = n;
  s.k = n - f;
  s.cur.assign(s.k, 0);
  s.best_diam = 0.0;
  s.have_best = false;
  s.rec(0, 0, 0.0);
  if (!s.have_best) return BM_EINVAL;
  for (int i = 0; i < s.k; ++i) sel_out[i] = s.best[i];
  return 0;
}
