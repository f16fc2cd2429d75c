// api.cpp — ABI bookkeeping and host-only entry points of libbm_gar.so.
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
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
Tuning& tuning_mutable() {
  static Tuning t = {env_int("BM_COL_BURST", 8),  env_int("BM_MEAN_BURST", 8),
                           env_int("BM_STEP_BURST", 8), env_int("BM_STEP_STREAM", 0), env_int("BM_PAIR_MODE", 0),
                           env_int("BM_PAIR_PLANES", 0), env_int("BM_PAIR_DITHER", 0), env_double("BM_PAIR_TAU", 2e-3),
                           env_int("BM_STUDY_BURST", 8), env_int("BM_STEP_STAGGER_US", 0),
                           env_int("BM_GRAM_STEADY", 1), env_int("BM_BULYAN_SHORT", 1),
                           env_int("BM_PAIR_LOAD_NT", 1), env_int("BM_RANK_ALGO", 0), env_int("BM_COL_WIDE", 1), env_int("BM_BRUTE_BUDGET", 0)};
  return t;
}
const Tuning& tuning() { return tuning_mutable(); }
}  // namespace bm

extern "C" int bm_abi_version(void) { return 23; }

// Launch-shape knobs of the A/B runs, settable inside one process (the environment is read once, at the first call):
// alternating two settings on the same data, on the same box, is the only comparison that resolves a 1 % effect.
extern "C" int bm_tuning_set(const char* name, int value) {
  using namespace bm;
  if (name == nullptr) return BM_EINVAL;
  Tuning& t = tuning_mutable();
  const struct { const char* key; int* slot; } knobs[] = {
      {"BM_COL_BURST", &t.col_burst}, {"BM_MEAN_BURST", &t.mean_burst}, {"BM_STEP_BURST", &t.step_burst},
      {"BM_STEP_STREAM", &t.step_stream}, {"BM_PAIR_MODE", &t.pair_mode}, {"BM_PAIR_PLANES", &t.pair_planes},
      {"BM_PAIR_DITHER", &t.pair_dither}, {"BM_STUDY_BURST", &t.study_burst},
      {"BM_STEP_STAGGER_US", &t.step_stagger_us}, {"BM_GRAM_STEADY", &t.gram_steady},
      {"BM_BULYAN_SHORT", &t.bulyan_short}, {"BM_PAIR_LOAD_NT", &t.pair_load_nt},
      {"BM_RANK_ALGO", &t.rank_algo}, {"BM_COL_WIDE", &t.col_wide}, {"BM_BRUTE_BUDGET", &t.brute_budget}};
  for (const auto& k : knobs)
    if (strcmp(name, k.key) == 0) {
      *k.slot = value;
      return 0;
    }
  return BM_EINVAL;
}

extern "C" const char* bm_error_string(int code) {
  if (code == 0) return "success";
  if (code == BM_EINVAL) return "invalid argument for libbm_gar";
  if (code == BM_ENOCOMM) return "RCCL is not available in this process (librccl.so.1 could not be bound)";
  if (code == BM_ECOMM) return "an RCCL call failed";
  if (code < 0) return hipGetErrorString(static_cast<hipError_t>(-code));
  return "unknown libbm_gar status";
}

// Host-side subset search of the Brute rule (aggregators/brute.py:47-68): among the C(n, n-f) subsets of n-f rows,
// visited in lexicographic order, the FIRST one of strictly smallest diameter (largest pairwise distance, at least 0);
// subsets that touch a non-finite distance are never candidates.
//
// The reference enumerates every subset (53 130 at n = 25, f = 5; 1.6e11 at n = 51, f = 12, which it cannot finish).
// The same answer comes from two questions about the graph G(t) = {pairs at finite distance <= t}:
//   1. the smallest t among the distances (and 0) for which G(t) holds n-f mutually adjacent rows — by bisection over
//      the sorted distances, G(t) only grows with t;
//   2. the lexicographically first such set in G(t*): position by position, the smallest row that still extends to one.
// "n-f mutually adjacent rows among `cand`" is "at most |cand| - (n-f) rows removed so that no non-adjacent pair is
// left", a search tree of depth f at most (take the row with the most non-neighbours: either it goes, or all of them
// go): 2^f leaves in the worst case instead of C(n, f) subsets, microseconds at both shapes above.
// Rows are bit sets (n <= BM_MAX_ROWS = 64).  Like the reference's loop, only dist[x*n + y] with x < y is read.
namespace {
struct BruteGraph {
  int n;
  uint64_t adj[BM_MAX_ROWS];  // adj[i]: rows j != i with a finite distance <= the current threshold

  void build(const double* dist, double t) {
    for (int i = 0; i < n; ++i) adj[i] = 0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) {
        const double v = dist[i * n + j];
        if (isfinite(v) && v <= t) {
          adj[i] |= (uint64_t)1 << j;
          adj[j] |= (uint64_t)1 << i;
        }
      }
  }

  // are there `need` mutually adjacent rows among `cand`?
  bool has_clique(uint64_t cand, int need) const {
    const int count = __builtin_popcountll(cand);
    if (count < need) return false;
    if (need <= 1) return true;
    int worst = -1, worst_missing = 0;
    for (uint64_t rest = cand; rest != 0; rest &= rest - 1) {
      const int u = __builtin_ctzll(rest);
      const int missing = __builtin_popcountll(cand & ~adj[u] & ~((uint64_t)1 << u));
      if (missing > worst_missing) {
        worst = u;
        worst_missing = missing;
      }
    }
    if (worst < 0) return true;  // no non-adjacent pair left: `cand` itself is one, of count >= need rows
    const int budget = count - need;
    if (budget == 0) return false;
    const uint64_t bit = (uint64_t)1 << worst;
    if (worst_missing <= budget && has_clique(cand & (adj[worst] | bit), need)) return true;  // it stays, they go
    return has_clique(cand & ~bit, need);                                                      // it goes
  }
};
}  // namespace

extern "C" int bm_brute_select(const double* dist_nxn, int n, int f, int32_t* sel_out) {
  if (dist_nxn == nullptr || sel_out == nullptr || n < 1 || n > BM_MAX_ROWS || f < 0 || n - f < 1)
    return BM_EINVAL;
  const int k = n - f;
  const uint64_t everyone = n == 64 ? ~(uint64_t)0 : (((uint64_t)1 << n) - 1);
  // candidate diameters: 0 (the reference's running maximum starts there) and every finite distance, ascending
  std::vector<double> values;
  values.reserve((size_t)n * (n - 1) / 2 + 1);
  values.push_back(0.0);
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const double v = dist_nxn[i * n + j];
      if (isfinite(v) && v > 0.0) values.push_back(v);
    }
  std::sort(values.begin(), values.end());
  values.erase(std::unique(values.begin(), values.end()), values.end());
  BruteGraph g;
  g.n = n;
  g.build(dist_nxn, values.back());
  if (!g.has_clique(everyone, k)) return BM_EINVAL;  // every subset touches a non-finite distance
  size_t lo = 0, hi = values.size() - 1;             // smallest index whose graph holds k mutually adjacent rows
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    g.build(dist_nxn, values[mid]);
    if (g.has_clique(everyone, k))
      hi = mid;
    else
      lo = mid + 1;
  }
  g.build(dist_nxn, values[lo]);
  // the first subset in lexicographic order: the smallest row that still leaves a completion among the rows above it
  uint64_t cand = everyone;
  int chosen = 0;
  for (int c = 0; c < n && chosen < k; ++c) {
    const uint64_t bit = (uint64_t)1 << c;
    if ((cand & bit) == 0) continue;
    const uint64_t above = c == 63 ? 0 : ~(((uint64_t)1 << (c + 1)) - 1);
    const uint64_t next = cand & g.adj[c] & above;
    if (g.has_clique(next, k - chosen - 1)) {
      sel_out[chosen++] = c;
      cand = next;
    }
  }
  return chosen == k ? 0 : BM_EINVAL;
}
