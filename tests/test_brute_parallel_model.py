"""The device Brute search (csrc/brute.hip: kBruteWaves waves ask "does G(t) hold n - f mutually adjacent rows" at
once) as a Python model of its three phases — pivots by row-major position among the open candidates, bounds closing in
on the tightest answers of a round, extraction of the lexicographically first set W rows at a time with the "what is
left is the rest" shortcut — against the host search bm_brute_select, which tests/test_host_logic.py pins on
exhaustive enumeration of aggregators/brute.py:47-68.  The HIP code itself is compared with the host search on the
GPU (tests/test_gpu_parity_r4.py); this file is what says the ALGORITHM is the reference's, without a GPU."""

import itertools
import math

import numpy as np
import pytest

from tests.test_host_logic import _brute_select

W = 16


def _graph(dist, n, t):
  adj = [0] * n
  for i in range(n):
    for j in range(n):
      v = dist[i, j] if i < j else dist[j, i]
      if j != i and math.isfinite(v) and v <= t:
        adj[i] |= 1 << j
  return adj


def _has_clique(adj, cand, need):
  """The search tree of BruteWave::has_clique: forced removals, then the row with the most non-neighbours."""
  stack, have = [], True
  while True:
    if not have:
      if not stack:
        return False
      cand = stack.pop()
    have = False
    count = bin(cand).count("1")
    if count < need:
      continue
    if need <= 1:
      return True
    missing = {i: bin(cand & ~adj[i] & ~(1 << i)).count("1") for i in range(len(adj)) if cand >> i & 1}
    if not any(missing.values()):
      return True
    budget = count - need
    if budget == 0:
      continue
    forced = sum(1 << i for i, m in missing.items() if m > budget)
    if forced:
      cand &= ~forced
      have = True
      continue
    worst = max(missing, key=lambda i: (missing[i], -i))
    goes = cand & ~(1 << worst)
    if missing[worst] <= budget:
      stack.append(goes)
      cand &= adj[worst] | (1 << worst)
    else:
      cand = goes
    have = True


def parallel_search(dist, n, f, waves=W):
  """(status, selection, rounds of phase 2, rounds of phase 3) as brute_select_kernel computes them."""
  k = n - f
  everyone = (1 << n) - 1
  upper = [dist[i, j] for i in range(n) for j in range(i + 1, n)]  # row-major, pairs i < j
  finite = [v for v in upper if math.isfinite(v)]
  vmax = max([0.0] + finite)
  if not _has_clique(_graph(dist, n, vmax), everyone, k):
    return -1, None, 0, 0
  lo, hi, rounds = 0.0, vmax, 0
  if _has_clique(_graph(dist, n, 0.0), everyone, k):
    hi = 0.0
  else:
    while True:
      open_ = [v for v in upper if math.isfinite(v) and lo < v < hi]
      total = len(open_)
      if total == 0:
        break
      rounds += 1
      askers = min(total, waves)
      answers = []
      for w in range(askers):
        pivot = open_[((2 * w + 1) * total) // (2 * askers)]
        answers.append((pivot, _has_clique(_graph(dist, n, pivot), everyone, k)))
      for pivot, yes in answers:
        if yes and pivot < hi:
          hi = pivot
        if not yes and pivot > lo:
          lo = pivot
  adj = _graph(dist, n, hi)
  # 3a. the rows that lie in SOME set of k mutually adjacent rows (n independent questions, `waves` at a time): no other
  #     row can be chosen at any position, and when exactly k rows are left they are the answer
  core = sum(1 << c for c in range(n) if _has_clique(adj, adj[c], k - 1))
  rounds3 = -(-n // waves)
  if bin(core).count("1") == k:
    return 0, [i for i in range(n) if core >> i & 1], rounds, rounds3
  # 3b. position by position among the rows of the core; a round tries the prefixes c_0, c_0 c_1, ... of the lowest
  #     open rows (wave v assumes c_0 .. c_{v-1} chosen): the longest prefix that extends is taken whole, and the row
  #     behind it has then failed exactly the question the sequential search would have asked
  cand, skipped, chosen, sel = core, 0, 0, []
  while chosen < k:
    open_rows = [i for i in range(n) if (cand & ~skipped) >> i & 1]
    if bin(cand).count("1") == k - chosen and skipped == 0:
      sel += [i for i in range(n) if cand >> i & 1]
      chosen = k
      break
    if not open_rows:
      return -1, None, rounds, rounds3
    rounds3 += 1
    tried = open_rows[:waves]
    accepted, state = 0, cand
    for v, c in enumerate(tried):
      if chosen + v + 1 > k:
        break
      nxt = state & adj[c] & ~((1 << (c + 1)) - 1)
      if not (state >> c & 1) or not _has_clique(adj, nxt, k - chosen - v - 1):  # (c must still be in play)
        break  # (answers are monotone in v: the kernel takes the longest run of "yes" from v = 0)
      state, accepted = nxt, v + 1
    sel += tried[:accepted]
    chosen += accepted
    if accepted > 0:
      cand, skipped = state, 0
    if accepted < len(tried) and chosen < k:
      skipped |= 1 << tried[accepted]   # it failed with exactly the prefix the sequential search would have had
  return 0, sel, rounds, rounds3


def _lattice(rng, n, spread=6, bad_rows=0):
  pts = rng.integers(0, spread, size=(n, 2)).astype(np.float64)  # many exact ties
  dist = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
  for bad in rng.choice(n, size=bad_rows, replace=False):
    dist[bad, :] = dist[:, bad] = math.nan
    dist[bad, bad] = 0
  return dist


def test_parallel_search_is_the_host_search_on_small_tied_matrices():
  rng = np.random.default_rng(11)
  for n, f in ((4, 1), (6, 1), (8, 3), (10, 2), (12, 5), (13, 4)):
    for trial in range(25):
      dist = _lattice(rng, n, bad_rows=(trial % 5 == 0) + (trial % 10 == 0) * f)
      rc, want = _brute_select(dist, n, f)
      status, got, _, _ = parallel_search(dist, n, f)
      assert (status == 0) == (rc == 0), (n, f, trial)
      if rc == 0:
        assert got == want, (n, f, trial)


@pytest.mark.parametrize("n,f", [(25, 5), (25, 11), (51, 12), (64, 20)])
def test_parallel_search_at_the_reference_shapes(n, f):
  """n = 25 / 51 (reproduce.py:122-209): continuous distances (no ties), clustered points (the honest cluster + far
  outliers of an attack), and lattice points (ties everywhere); the rounds stay few."""
  rng = np.random.default_rng(5)
  worst2 = worst3 = 0
  for trial in range(6):
    if trial % 3 == 0:
      pts = rng.normal(size=(n, 8))
    elif trial % 3 == 1:
      pts = np.concatenate([rng.normal(size=(n - f, 8)), 6.0 + 0.01 * rng.normal(size=(f, 8))])
      pts = pts[rng.permutation(n)]
    else:
      pts = rng.integers(0, 4, size=(n, 3)).astype(np.float64)
    dist = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    rc, want = _brute_select(dist, n, f)
    status, got, r2, r3 = parallel_search(dist, n, f)
    assert rc == 0 and status == 0 and got == want, (n, f, trial)
    worst2, worst3 = max(worst2, r2), max(worst3, r3)
  assert worst2 <= 6 and worst3 <= 4 + 2 * f + 3, (worst2, worst3)


def test_parallel_search_with_fewer_open_candidates_than_waves():
  """Three rows: one or two open candidates per round, every wave beyond them asks nothing."""
  dist = np.array([[0., 1., 3.], [1., 0., 2.], [3., 2., 0.]])
  for f in (0, 1):
    rc, want = _brute_select(dist, 3, f)
    assert rc == 0 and parallel_search(dist, 3, f)[1] == want
  dist[0, 1] = dist[1, 0] = math.inf
  rc, want = _brute_select(dist, 3, 1)
  assert rc == 0 and parallel_search(dist, 3, 1)[1] == want == [1, 2]
