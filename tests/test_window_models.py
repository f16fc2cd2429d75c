"""Bulyan pass 2, the choice of the beta values closest to the median (aggregators/bulyan.py:79-82: `topk` of the
smallest |selected - median| per column) as csrc/bulyan.hip makes it on a SORTED column — a model of its two forms:

  long  : s = 1 + the last t in [0, theta-beta) with |sel[t] - med| > |sel[t+beta] - med|  (0 if none), window [s, s+beta);
  short : the same search restricted to the t whose window straddles the median (BM_BULYAN_SHORT, the default).

Checked here without a GPU, on every Bulyan shape the reference runs (and a few more) and on columns with ties,
repeated medians and infinite values: (1) the two forms select windows holding the same values, and their fp32 sums
— masked accumulation in index order, as the kernel writes it — have the same bits; (2) the window is a valid answer
of the reference's `topk`: no value outside it is strictly closer to the median than a value inside it."""

import numpy as np
import pytest

SHAPES = [(11, 2), (15, 3), (19, 2), (19, 4), (25, 5), (25, 3), (35, 5), (35, 8), (51, 12), (51, 5), (63, 15), (64, 1)]


def _windows(sel, theta, beta):
  """(start of the long form, start of the short form, fp32 sums of both) for one sorted column."""
  med_i = (theta - 1) // 2
  med = sel[med_i]
  s_long = 0
  for t in range(theta - beta):
    if abs(sel[t] - med) > abs(sel[t + beta] - med):
      s_long = t + 1
  lo = max(med_i - beta + 1, 0)
  tend = min(med_i, theta - beta)
  iend = min(med_i + beta, theta)
  s_short = lo
  for t in range(lo, tend):
    if abs(sel[t] - med) > abs(sel[t + beta] - med):
      s_short = t + 1
  w_long = np.float32(0.0)
  for i in range(theta):
    w_long = np.float32(w_long + (sel[i] if s_long <= i < s_long + beta else np.float32(0.0)))
  w_short = np.float32(0.0)
  for i in range(lo, iend):
    w_short = np.float32(w_short + (sel[i] if s_short <= i < s_short + beta else np.float32(0.0)))
  return s_long, s_short, w_long, w_short


def _columns(rng, theta, count):
  kinds = []
  x = rng.standard_normal((count, theta)).astype(np.float32)
  kinds.append(x)
  kinds.append((np.round(x * 2) / 2).astype(np.float32))                          # ties everywhere
  y = x.copy()
  y[:, theta // 3: 2 * theta // 3 + 1] = y[:, [theta // 2]]                        # the median repeated
  kinds.append(y)
  z = x.copy()
  z[:, :2] = -np.inf                                                             # infinite values at the ends (finite median)
  z[:, -1:] = np.inf
  kinds.append(z)
  kinds.append(np.where(rng.random((count, theta)) < 0.5, np.float32(-0.0), np.float32(0.0)))  # zeros of both signs
  kinds.append(rng.integers(-2, 3, (count, theta)).astype(np.float32))             # few distinct values
  return np.sort(np.concatenate(kinds), axis=1)


@pytest.mark.parametrize("n,f", SHAPES)
def test_short_window_search_selects_what_the_long_one_selects(n, f):
  theta, beta = n - 2 * f - 2, n - 4 * f - 2
  assert beta >= 1
  rng = np.random.default_rng(100 * n + f)
  cols = _columns(rng, theta, 300)
  med_i = (theta - 1) // 2
  with np.errstate(invalid="ignore"):
    for sel in cols:
      if not np.isfinite(sel[med_i]):
        continue  # (a wave that holds such a column takes the long form)
      s_long, s_short, w_long, w_short = _windows(sel, theta, beta)
      assert np.array_equal(sel[s_long:s_long + beta], sel[s_short:s_short + beta]), (n, f, sel, s_long, s_short)
      assert w_long.tobytes() == w_short.tobytes(), (n, f, sel)
      # a valid `topk` of the smallest deviations: nothing outside the window is strictly closer to the median
      dev = np.abs(sel - sel[med_i])
      inside = dev[s_long:s_long + beta]
      outside = np.concatenate([dev[:s_long], dev[s_long + beta:]])
      if outside.size:
        assert inside.max() <= outside.min(), (n, f, sel, s_long)


@pytest.mark.parametrize("n,f", [(5, 1), (11, 2), (11, 4), (25, 5), (25, 11), (51, 12), (51, 24), (64, 20), (64, 31)])
def test_closest_to_centre_window_of_phocas_and_meamed_is_a_valid_topk(n, f):
  """csrc/colwise_kernels.h, column_rule<PHOCAS|MEAMED>: on the sorted column the n - f values closest to the centre c
  (trimmed mean or median: aggregators/trmean.py:35-50, `topk` of the smallest |g - c|) are the window [s, s + m) with
  s = 1 + the last t < f such that |x[t] - c| > |x[t + m] - c|.  Model check: nothing outside that window is strictly
  closer to c than something inside it — for centres that are column values (median), means of a part of the column
  (trimmed mean) and arbitrary numbers, with ties and infinite values."""
  m = n - f
  rng = np.random.default_rng(7 * n + f)
  cols = _columns(rng, n, 200)
  with np.errstate(invalid="ignore", over="ignore"):
    for x in cols:
      centres = [x[(n - 1) // 2], np.float32(x[f:n - f].astype(np.float64).mean()), np.float32(rng.standard_normal())]
      for c in centres:
        if not np.isfinite(c):
          continue
        s = 0
        for t in range(f):
          if abs(x[t] - c) > abs(x[t + m] - c):
            s = t + 1
        dev = np.abs(x - c)
        inside, outside = dev[s:s + m], np.concatenate([dev[:s], dev[s + m:]])
        assert 0 <= s <= f and inside.size == m
        if outside.size:
          assert inside.max() <= outside.min(), (n, f, c, x, s)


# ---------------------------------------------------------------------------- #
# phocas / meamed (aggregators/trmean.py:35-50): the n - f values closest to the centre, as csrc/colwise_kernels.h
# (closest_sum_static) sums them on registers with static indices.

def _closest_forms(x, f, c):
  """(window start, fp32 sum walking from s — the form of rounds 2-5 —, fp32 sum of the static-index form) of a sorted column."""
  n = len(x)
  m = n - f
  s = 0
  for t in range(f):
    if abs(np.float32(x[t] - c)) > abs(np.float32(x[t + m] - c)):
      s = t + 1
  walk = np.float32(0.0)
  for i in range(m):
    walk = np.float32(walk + x[s + i])
  static = np.float32(0.0)
  for i in range(n):
    if i < f:
      static = np.float32(static + (x[i] if i >= s else np.float32(0.0)))
    elif i >= m:
      static = np.float32(static + (x[i] if i < s + m else np.float32(0.0)))
    else:
      static = np.float32(static + x[i])
  return s, walk, static


@pytest.mark.parametrize("n,f", [(3, 1), (7, 1), (11, 2), (11, 5), (25, 5), (25, 11), (25, 12), (51, 12), (51, 25), (64, 31), (9, 0)])
def test_closest_to_centre_sum_on_static_indices_has_the_bits_of_the_walk(n, f):
  """The ranks f .. n-f-1 lie in every possible window [s, s + n - f), 0 <= s <= f, and are added unconditionally; the 2f
  ranks at the ends are added as `x or +0`.  Same bits as walking the window from s — on random columns, columns with
  ties and repeated values, negative zeros, infinities (NaN replaced by +inf sorts last), and for every centre the
  two rules use (the median, the trimmed mean)."""
  rng = np.random.default_rng(n * 100 + f)
  for trial in range(400):
    x = rng.standard_normal(n).astype(np.float32)
    kind = trial % 5
    if kind == 1:
      x = np.round(x)                      # many ties
    elif kind == 2:
      x[rng.integers(0, n, size=max(1, n // 4))] = np.float32(np.inf)
    elif kind == 3:
      x[:] = x[0]                          # one value
    elif kind == 4:
      x[rng.integers(0, n)] = np.float32(-0.0)
    x = np.sort(x)
    for c in (x[(n - 1) // 2], np.float32(x[f:n - f].astype(np.float64).mean()) if n - 2 * f > 0 else x[0]):
      with np.errstate(invalid="ignore"):  # (inf - inf: the comparison is then false, as on the device)
        s, walk, static = _closest_forms(x, f, np.float32(c))
      assert 0 <= s <= f
      assert walk.tobytes() == static.tobytes() or (np.isnan(walk) and np.isnan(static)), (n, f, trial, s, walk, static)
