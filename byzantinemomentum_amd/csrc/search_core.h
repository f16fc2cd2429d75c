// search_core.h — the scalar pieces of the attacks' factor search (attacks/identical.py:67-77, tools/misc.py:468-514)
// shared by the host form (linesearch.cpp) and the device form (search_device.hip): the cursor of the exploration and
// the closed forms of a candidate's distances and objective.  One definition, compiled for both sides (the whole
// library is built with -ffp-contract=off), so that the two forms propose the same candidates and compare the same
// bits: the device form is tested bit for bit against the host form.
#pragma once
#include "bm_common.h"

namespace bm {

// The exploration of tools/misc.py:468-514 as a CURSOR: propose() names the next abscissa, report() takes the value
// measured there.  Behaviour to reproduce (the candidates must be the reference's, evaluation for evaluation):
//   GROW    probe = incumbent + step; a strictly better value moves the incumbent there and doubles the step,
//           the first value that is not better multiplies the step by `ratio` and ends the phase;
//   SHRINK  the probe walks towards the incumbent and oscillates around it (+step while left of it, else
//           -step, folded back into x >= 0 by repeated halving of the overshoot); the step is multiplied by
//           `ratio` after every evaluation; strictly better values move the incumbent.
enum { kSearchFirst = 0, kSearchGrow = 1, kSearchShrink = 2 };

__host__ __device__ inline void cursor_begin(bm_search* c, double start, double delta, double ratio) {
  c->best_x = start;
  c->best_y = 0.0;
  c->probe = start;
  c->step = delta;
  c->ratio = ratio;
  c->phase = kSearchFirst;
  c->evaluations = 0;
  c->awaiting = 0;
  c->reserved = 0;
}

__host__ __device__ inline void cursor_propose(bm_search* c) {
  switch (c->phase) {
    case kSearchFirst:
      break;  // probe already holds the starting point
    case kSearchGrow:
      c->probe = c->best_x + c->step;
      break;
    default:
      if (c->probe < c->best_x) {
        c->probe += c->step;
      } else {
        double x = c->probe - c->step;
        while (x < 0.0) x = 0.5 * (x + c->probe);
        c->probe = x;
      }
  }
}

__host__ __device__ inline void cursor_report(bm_search* c, double y) {
  const bool better = (c->phase == kSearchFirst) || (y > c->best_y);  // strict: equal values never move the incumbent
  if (better) {
    c->best_x = c->probe;
    c->best_y = y;
  }
  switch (c->phase) {
    case kSearchFirst:
      c->phase = kSearchGrow;
      break;
    case kSearchGrow:
      if (better) {
        c->step *= 2.0;
      } else {
        c->step *= c->ratio;
        c->phase = kSearchShrink;
      }
      break;
    default:
      c->step *= c->ratio;
  }
  ++c->evaluations;
}

// With u_i = h_i - avg and the Byzantine row of candidate t being avg + t * att, from the (h+2) x (h+2) squared
// distances `ext` among {h_1..h_h, avg, avg + att} (e = h + 2 its row length):
//   a_i = |u_i|^2 = ext[i][h]        c = |att|^2 = ext[h][h+1]
//   w_i = <u_i, att>   = (a_i + c - ext[i][h+1]) / 2
//   <u_i, u_j>         = (a_i + a_j - ext[i][j]) / 2
//   |h_i - byz(t)|^2   = a_i - 2 t w_i + t^2 c      (clamped at 0: rounding of a candidate that coincides with a row)
__host__ __device__ inline double attack_w(double a_i, double c, double ext_i_att) { return 0.5 * (a_i + c - ext_i_att); }
__host__ __device__ inline double attack_uu(double a_i, double a_j, double ext_ij, bool same) {
  return same ? a_i : 0.5 * (a_i + a_j - ext_ij);
}
__host__ __device__ inline double attack_candidate_sq(double a_i, double w_i, double c, double t) {
  const double q = a_i - 2.0 * t * w_i + t * t * c;
  return (q < 0.0) ? 0.0 : q;
}
// |mean(selected rows) - avg|^2 = |sum_{i in S} u_i + kb t att|^2 / count^2 with S the selected honest rows and kb
// the selected Byzantine copies: quad = sum_{i in S} row_i, row_i = sum_{j in S} <u_i, u_j>, lin = sum_{i in S} w_i, in
// a FIXED order so that the value depends on the selected SET only (two candidates that select the same honest rows
// and no Byzantine one compare equal, as they do in the reference where the rule then returns the same vector): the
// inner sums as (first half of the columns 0 .. attack_row_span(h) - 1 in index order) + (second half in index order),
// a column adding its value when it is selected and 0.0 when it is not or lies beyond h (the device reads a row in
// groups and adds without a branch, two chains in flight); the outer ones over 64 slots (row i in slot i, 0.0 in the slots of unselected rows) in the
// order a wave can follow with all its lanes at once — butterfly_order_sum below (a sequential outer sum costs the
// device a chain of ~40 dependent fp64 additions of 13 ns each per candidate, profiles/r06_device_search.txt).
__host__ __device__ inline int attack_row_span(int h) { return (h + 7) & ~7; }
// slot[i] <- slot[i] + slot[i ^ 1], then ^ 2, ^ 4, ..., ^ 32 (all 64 slots at every level): slot[0] at the end.  The device
// form is butterfly_sum (search_device.hip).
inline double butterfly_order_sum(const double (&slots)[BM_MAX_ROWS]) {
  double cur[BM_MAX_ROWS], nxt[BM_MAX_ROWS];
  for (int i = 0; i < BM_MAX_ROWS; ++i) cur[i] = slots[i];
  for (int s = 1; s < BM_MAX_ROWS; s <<= 1) {
    for (int i = 0; i < BM_MAX_ROWS; ++i) nxt[i] = cur[i] + cur[i ^ s];
    for (int i = 0; i < BM_MAX_ROWS; ++i) cur[i] = nxt[i];
  }
  return cur[0];
}

__host__ __device__ inline double attack_objective_value(double quad, double lin, int kb, double t, double c, int count) {
  const double cnt = (double)count;
  return (quad + 2.0 * kb * t * lin + (double)kb * kb * t * t * c) / (cnt * cnt);
}

}  // namespace bm
