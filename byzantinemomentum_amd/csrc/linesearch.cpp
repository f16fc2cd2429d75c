// Line-searching attacks on the host: attacks/identical.py:67-77 (the factor search of the "empire",
// "little" and "bulyan" attacks, the DEFAULT of the reference: factor=-16); the exploration it asks
// tools/misc.py:468-514 for is offered as a caller-driven cursor (bm_search_*).
//
// The reference evaluates the aggregation rule on n d-sized vectors once per candidate factor (16 times
// per step).  For the rules whose output is the mean of a selected subset (Multi-Krum, Brute, Average),
// every quantity the search needs is a function of the inner products among
//     u_i = h_i - avg (the h honest rows around their mean)   and   att (the attack direction),
// because the Byzantine row of candidate t is avg + t * att:
//     |h_i - byz(t)|^2 = |u_i|^2 - 2 t <u_i, att> + t^2 |att|^2          (the rule's distance matrix)
//     GAR(t) - avg     = (sum_{i in S} u_i + kb * t * att) / M            (S: selected honest rows,
//                                                                          kb: selected Byzantine copies)
// Those inner products come from ONE squared-distance pass over the h + 2 rows {h_1..h_h, avg, avg+att}
// (bm_pairwise_sqdist), after which a candidate costs O(n^2 log n) host flops and no device work at all.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "bm_common.h"
#include "search_core.h"  // the cursor and the closed forms, shared with the device form (search_device.hip)

extern "C" int bm_brute_select(const double* dist_nxn, int n, int f, int32_t* sel_out);

namespace {

using namespace bm;

struct AttackGeometry {
  int h, k, n;
  std::vector<double> a;    // |u_i|^2
  std::vector<double> w;    // <u_i, att>
  std::vector<double> uu;   // <u_i, u_j>, h x h
  std::vector<double> hh;   // |h_i - h_j|^2, h x h
  double c;                 // |att|^2

  AttackGeometry(const double* ext, int h_, int k_) : h(h_), k(k_), n(h_ + k_), a(h_), w(h_), uu((size_t)h_ * h_), hh((size_t)h_ * h_) {
    const int e = h + 2;  // ext is (h+2) x (h+2): rows 0..h-1 honest, h = avg, h+1 = avg + att
    c = ext[h * e + h + 1];
    for (int i = 0; i < h; ++i) {
      a[i] = ext[i * e + h];
      w[i] = attack_w(a[i], c, ext[i * e + h + 1]);
    }
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < h; ++j) {
        hh[(size_t)i * h + j] = ext[i * e + j];
        uu[(size_t)i * h + j] = attack_uu(a[i], a[j], ext[i * e + j], i == j);
      }
  }

  // n x n squared distances of honests + [avg + t*att] * k
  void sqdist(double t, std::vector<double>& sq) const {
    sq.assign((size_t)n * n, 0.0);
    for (int i = 0; i < h; ++i) {
      for (int j = 0; j < h; ++j) sq[(size_t)i * n + j] = hh[(size_t)i * h + j];
      const double q = attack_candidate_sq(a[i], w[i], c, t);
      for (int j = h; j < n; ++j) {
        sq[(size_t)i * n + j] = q;
        sq[(size_t)j * n + i] = q;
      }
    }
  }

  // |mean(selected rows) - avg|^2 (search_core.h: row sums over the two halves of the row's span in index order, every
  // column adding its value or 0.0; then the butterfly order over 64 slots)
  double objective(const std::vector<int>& sel_sorted, double t) const {
    int kb = 0;
    bool on[BM_MAX_ROWS] = {false};
    for (int i : sel_sorted) {
      if (i >= h) ++kb;
      else on[i] = true;
    }
    double rows[BM_MAX_ROWS] = {0.0}, ws[BM_MAX_ROWS] = {0.0};
    const int span = attack_row_span(h);
    for (int i = 0; i < h; ++i) {
      if (!on[i]) continue;
      ws[i] = w[i];
      double lo = 0.0, hi = 0.0;
      for (int j = 0; j < span / 2; ++j) lo += (j < h && on[j]) ? uu[(size_t)i * h + j] : 0.0;
      for (int j = span / 2; j < span; ++j) hi += (j < h && on[j]) ? uu[(size_t)i * h + j] : 0.0;
      rows[i] = lo + hi;
    }
    return attack_objective_value(butterfly_order_sum(rows), butterfly_order_sum(ws), kb, t, c, (int)sel_sorted.size());
  }
};

// Multi-Krum / Bulyan ranking on the host with the semantics of krum_rank_kernel (pairwise.hip) / krum.py:44-62,
// bulyan.py:48-62: distances = sqrt, non-finite -> +inf; score = ascending fp64 sum of the `take` smallest of the
// row; stable order of the scores (ties to the lower index).
// `take`: distances summed per row — n-f-1 for Krum (krum.py:59-60), m for Bulyan (bulyan.py:56-62)
void rank_order(const std::vector<double>& sq, int n, int take, std::vector<int>& order) {
  std::vector<double> score(n), row;
  take = std::max(0, std::min(take, n - 1));
  for (int i = 0; i < n; ++i) {
    row.clear();
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      double v = std::sqrt(sq[(size_t)i * n + j]);
      if (!std::isfinite(v)) v = INFINITY;
      row.push_back(v);
    }
    std::sort(row.begin(), row.end());
    double s = 0.0;
    for (int t = 0; t < take; ++t) s += row[t];
    score[i] = s;
  }
  order.resize(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return score[x] < score[y]; });
}

int select(const AttackGeometry& g, int f, int rule, int m, double t, std::vector<int>& sel) {
  const int n = g.n;
  std::vector<double> sq;
  if (rule == BM_RULE_AVERAGE) {
    sel.resize(n);
    for (int i = 0; i < n; ++i) sel[i] = i;
    return 0;
  }
  g.sqdist(t, sq);
  if (rule == BM_RULE_KRUM) {
    std::vector<int> order;
    rank_order(sq, n, n - f - 1, order);
    sel.assign(order.begin(), order.begin() + m);
    return 0;
  }
  // brute.py:44-68 works on the distances themselves
  for (double& v : sq) v = std::sqrt(v);
  std::vector<int32_t> out(n - f);
  const int rc = bm_brute_select(sq.data(), n, f, out.data());
  if (rc != 0) return rc;
  sel.assign(out.begin(), out.end());
  return 0;
}

bool valid(const double* ext, int h, int k, int f, int rule, int& m) {
  const int n = h + k;
  if (ext == nullptr || h < 1 || k < 0 || n > BM_MAX_ROWS || f < 0) return false;
  if (rule == BM_RULE_KRUM) {
    if (m <= 0) m = n - f - 2;
    return m >= 1 && m <= n;
  }
  if (rule == BM_RULE_BRUTE) return n - f >= 1;
  return rule == BM_RULE_AVERAGE;
}

}  // namespace

extern "C" int bm_search_begin(bm_search* c, double start, double delta, double ratio) {
  if (c == nullptr || !(start >= 0.0) || !(delta > 0.0) || !(ratio > 0.5 && ratio < 1.0)) return BM_EINVAL;
  cursor_begin(c, start, delta, ratio);
  return 0;
}

extern "C" int bm_search_propose(bm_search* c, double* x_out) {
  if (c == nullptr || x_out == nullptr || c->awaiting) return BM_EINVAL;  // one report per proposal
  cursor_propose(c);
  c->awaiting = 1;
  *x_out = c->probe;
  return 0;
}

extern "C" int bm_search_report(bm_search* c, double y) {
  if (c == nullptr || !c->awaiting) return BM_EINVAL;
  cursor_report(c, y);
  c->awaiting = 0;
  return 0;
}

extern "C" int bm_attack_objective(const double* ext, int h, int k, int f, int rule, int m, double t,
                                   double* y_out, int32_t* sel_out, int32_t* count_out) {
  if (!valid(ext, h, k, f, rule, m) || y_out == nullptr) return BM_EINVAL;
  const AttackGeometry g(ext, h, k);
  std::vector<int> sel;
  const int rc = select(g, f, rule, m, t, sel);
  if (rc != 0) return rc;
  if (sel_out != nullptr)
    for (size_t i = 0; i < sel.size(); ++i) sel_out[i] = sel[i];
  if (count_out != nullptr) *count_out = (int32_t)sel.size();
  std::sort(sel.begin(), sel.end());
  *y_out = g.objective(sel, t);
  return 0;
}

// The ranking bm_krum_rank would give for honests + [avg + t*att] * k, from the scalars alone: what the factor search
// needs of the distance pass when the rule's output is NOT a function of inner products (Bulyan: its second pass
// runs on the vectors, bulyan.py:64-84, but the ranking that feeds it is).  Indices >= h are the Byzantine copies.
extern "C" int bm_attack_ranking(const double* ext, int h, int k, int f, int mode, int m, double t,
                                 int32_t* order_out) {
  const int n = h + k;
  if (ext == nullptr || order_out == nullptr || h < 1 || k < 0 || n > BM_MAX_ROWS || f < 0 ||
      (mode != BM_RANK_KRUM && mode != BM_RANK_BULYAN))
    return BM_EINVAL;
  if (m <= 0) m = n - f - 2;
  if (m < 1 || m > n) return BM_EINVAL;
  const AttackGeometry g(ext, h, k);
  std::vector<double> sq;
  g.sqdist(t, sq);
  std::vector<int> order;
  rank_order(sq, n, mode == BM_RANK_KRUM ? n - f - 1 : m, order);
  for (int i = 0; i < n; ++i) order_out[i] = order[i];
  return 0;
}

extern "C" int bm_attack_line_search(const double* ext, int h, int k, int f, int rule, int m, int evals,
                                     int negative, double* factor_out, double* trace_out) {
  if (!valid(ext, h, k, f, rule, m) || factor_out == nullptr || evals < 1) return BM_EINVAL;
  const AttackGeometry g(ext, h, k);
  std::vector<int> sel;
  bm_search cur;
  int rc = bm_search_begin(&cur, 0.0, 1.0, 0.8);  // the attack's call: tools.line_maximize(eval_factor, evals=evals)
  for (int e = 0; e < evals && rc == 0; ++e) {
    double x = 0.0;
    rc = bm_search_propose(&cur, &x);
    if (rc != 0) break;
    const double t = negative ? -x : x;  // identical.py:70-71
    rc = select(g, f, rule, m, t, sel);
    if (rc != 0) break;
    std::sort(sel.begin(), sel.end());
    const double y = g.objective(sel, t);
    if (trace_out != nullptr) {
      trace_out[2 * e] = x;
      trace_out[2 * e + 1] = y;
    }
    rc = bm_search_report(&cur, y);
  }
  *factor_out = cur.best_x;
  return rc;
}
