"""Host-side logic that needs no GPU: input validation (loud failure instead of a CPU fallback),
the Brute subset search of the C ABI (pure host code), and the drop-in registration into the
reference's registry."""

import ctypes
import itertools
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import gar_oracle as O
from tests.golden_io import CASES, Golden


def test_cpu_tensors_are_refused_loudly():
  import byzantinemomentum_amd as bm
  g = [torch.randn(16) for _ in range(7)]
  for call in (lambda: bm.median(g), lambda: bm.trmean(g, 1), lambda: bm.krum(g, 1), lambda: bm.bulyan(g, 1),
               lambda: bm.brute(g, 1), lambda: bm.aksel(g, 1), lambda: bm.compute_avg_dev_max(g)):
    with pytest.raises(bm.gars.GarInputError, match="no CPU fallback"):
      call()


def test_product_never_imports_the_oracle():
  """Static check: nothing under byzantinemomentum_amd/ or native/ mentions the oracle package."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  for pkg in ("byzantinemomentum_amd", "native"):
    for dirpath, _, files in os.walk(os.path.join(root, pkg)):
      for name in files:
        if name.endswith(".py"):
          text = open(os.path.join(dirpath, name)).read()
          assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, name)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
  from byzantinemomentum_amd import _lib
  monkeypatch.setattr(_lib, "_lib", None)
  monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libbm_gar.so")
  with pytest.raises(_lib.NativeLibraryError, match="no CPU fallback"):
    _lib.load()


def _brute_select(dist, n, f):
  from byzantinemomentum_amd import _lib
  lib = _lib.load()
  dist = np.ascontiguousarray(dist, dtype=np.float64)
  sel = (ctypes.c_int32 * (n - f))()
  rc = lib.bm_brute_select(dist.ctypes.data_as(ctypes.c_void_p), n, f, ctypes.cast(sel, ctypes.c_void_p))
  return rc, list(sel)


@pytest.mark.parametrize("name", [c for c in CASES if Golden(c).has("brute_selection")])
def test_brute_select_matches_reference_fixture(name):
  g = Golden(name)
  dist = O.pairwise_distances(g.gradients, "f32", clamp_nonfinite=False)
  rc, sel = _brute_select(dist, g.n, g.f)
  assert rc == 0 and sel == g.array("brute_selection").tolist()


def test_brute_select_random_matrices_against_exhaustive_search():
  rng = np.random.default_rng(3)
  for n, f in ((6, 1), (8, 3), (10, 2), (12, 5)):
    for trial in range(20):
      pts = rng.integers(0, 6, size=(n, 2)).astype(np.float64)  # many exact ties
      dist = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
      if trial % 5 == 0:
        bad = rng.integers(0, n)
        dist[bad, :] = dist[:, bad] = math.nan
        dist[bad, bad] = 0
      best, best_d = None, None
      for sub in itertools.combinations(range(n), n - f):
        blk = dist[np.ix_(sub, sub)]
        if not np.isfinite(blk).all():
          continue
        dm = blk.max()
        if best is None or dm < best_d:
          best, best_d = list(sub), dm
      rc, sel = _brute_select(dist, n, f)
      if best is None:
        assert rc != 0
      else:
        assert rc == 0 and sel == best


def _reference_selection(dist, n, f):
  """The loop of aggregators/brute.py:47-68 on a distance matrix: running maximum from 0, subsets touching a non-finite
  distance dropped, first subset of strictly smallest diameter."""
  best, best_d = None, None
  for sub in itertools.combinations(range(n), n - f):
    diam, ok = 0., True
    for x, y in itertools.combinations(sub, 2):
      v = dist[x, y]
      if not math.isfinite(v):
        ok = False
        break
      if v > diam:
        diam = v
    if ok and (best is None or diam < best_d):
      best, best_d = list(sub), diam
  return best


def test_brute_select_every_small_shape_with_ties_zeros_and_non_finite_distances():
  """bm_brute_select answers from the threshold graphs of the distances (smallest diameter by bisection, then the
  lexicographically first subset) instead of enumerating subsets: every (n, f) up to n = 11 — f = 0 and n - f = 1
  included — on integer grids (exact ties), coincident points (zero distances), inf / NaN entries."""
  rng = np.random.default_rng(11)
  for trial in range(400):
    n = int(rng.integers(1, 12))
    f = int(rng.integers(0, n))
    kind = trial % 4
    if kind == 0:
      pts = rng.integers(0, 4, size=(n, 2)).astype(np.float64)
    elif kind == 2:
      pts = rng.integers(0, 2, size=(n, 1)).astype(np.float64)
    else:
      pts = rng.standard_normal((n, 3))
    dist = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
    if kind == 3:
      for _ in range(int(rng.integers(0, 4))):
        i, j = rng.integers(0, n, 2)
        if i != j:
          dist[i, j] = dist[j, i] = rng.choice([math.inf, math.nan])
    want = _reference_selection(dist, n, f)
    rc, sel = _brute_select(dist, n, f)
    if want is None:
      assert rc != 0, (n, f)
    else:
      assert rc == 0 and sel == want, (n, f, kind)


def _stack_distances(n, f, seed, d=2048):
  """fp64 distances of a bench-like stack: honest rows N(mu, sigma_i^2), f Byzantine rows around -0.1 x their mean."""
  rng = np.random.default_rng(seed)
  h = n - f
  mu = 0.1 * rng.standard_normal(d)
  rows = [mu + s * rng.standard_normal(d) for s in np.linspace(0.5, 1.5, h)]
  byz = -0.1 * np.mean(rows, axis=0)
  rows += [byz + 0.3 * rng.standard_normal(d) for _ in range(f)]
  x = np.array(rows)
  return np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))


@pytest.mark.parametrize("n,f,seed", [(25, 5, 0), (25, 5, 1), (25, 11, 2), (51, 12, 3), (51, 12, 4)])
def test_brute_select_is_the_references_answer_at_the_bench_shapes(n, f, seed):
  """The reference's own loop is the check at n = 25, f = 5 (53 130 subsets); beyond, the oracle's checker, which
  decides the three defining properties (finite, strictly smallest diameter, first in lexicographic order) with
  its own search."""
  dist = _stack_distances(n, f, seed)
  if seed % 2 == 1:  # exact ties: two aliased rows and a pair exactly as far apart as another one
    dist[:, n - 1] = dist[:, n - 2]
    dist[n - 1, :] = dist[n - 2, :]
    dist[n - 2, n - 1] = dist[n - 1, n - 2] = dist[n - 1, n - 1] = 0.0
  rc, sel = _brute_select(dist, n, f)
  assert rc == 0
  assert O.brute_selection_is_the_references(dist, f, sel)
  if (n, f) == (25, 5):
    assert sel == O.brute_selection_from_distances(dist, f)


@pytest.mark.parametrize("n,f", [(25, 5), (25, 11), (51, 12), (51, 24), (64, 31)])
def test_brute_select_at_sizes_the_reference_cannot_enumerate(n, f):
  """C(51, 12) = 1.6e11 subsets: the reference's loop never ends there.  A planted answer: n - f rows
  within a ball of diameter < 1, the other f at distance > 5 from everything (ties and the lexicographic rule are
  pinned on the small shapes above)."""
  rng = np.random.default_rng(n * 100 + f)
  k = n - f
  inside = sorted(rng.permutation(n)[:k].tolist())
  pts = np.zeros((n, 8))
  for i in range(n):
    if i in inside:
      v = rng.standard_normal(8)
      pts[i] = 0.45 * rng.random() * v / np.linalg.norm(v)
    else:
      v = rng.standard_normal(8)
      pts[i] = (6.0 + 3.0 * rng.random()) * v / np.linalg.norm(v) + 20.0 * (i + 1)
  dist = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1))
  rc, sel = _brute_select(dist, n, f)
  assert rc == 0 and sel == inside
  # non-finite rows beyond the budget: no subset is left
  dist[:, :f + 1] = math.nan
  dist[:f + 1, :] = math.nan
  assert _brute_select(dist, n, f)[0] != 0


@pytest.mark.reference
def test_reference_discovers_native_package():
  """`import aggregators` of the UNMODIFIED reference with the repo on PYTHONPATH registers the
  native-* rules (aggregators/krum.py:22-26,159-166 and our own calls to aggregators.register)."""
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  code = ("import aggregators, sys\n"
          "names = sorted(n for n in aggregators.gars if n.startswith('native-'))\n"
          "print(','.join(names))\n"
          "assert aggregators.gars['native-aksel'].influence is not None\n"
          "assert aggregators.gars['native-trmean'].check(gradients=[1], f=1) is not None\n")
  env = dict(os.environ, PYTHONPATH=os.pathsep.join([root, "/root/reference"]))
  out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp")
  assert out.returncode == 0, out.stderr
  names = out.stdout.strip().splitlines()[-1].split(",")
  assert names == ["native-aksel", "native-average", "native-brute", "native-bulyan", "native-cge", "native-krum",
                   "native-meamed", "native-median", "native-phocas", "native-trmean"]


def test_burst_form_index_map_covers_every_column_group_once():
  """The burst forms (colwise_burst_kernel, selected_mean_burst_kernel, bulyan_pass2_kernel<…, BURST>) walk the
  column groups as v = it * (grid * T) + block * T + lane, in phases of `slots` iterations whose results sit in LDS
  slot (it - p0) * T + lane until the barrier.  Replay that arithmetic on the host: every group below nv is produced
  exactly once, staged in a slot no other live result of the same lane occupies, and written from that slot."""
  def replay(nv, grid, threads, slots):
    span = grid * threads
    iters = (nv + span - 1) // span
    written = np.zeros(nv, dtype=np.int32)
    for block in range(grid):
      lanes = np.arange(threads)
      first = block * threads + lanes
      p0 = 0
      while p0 < iters:
        p1 = min(p0 + slots, iters)
        stage = {}
        for it in range(p0, p1):
          v = it * span + first
          ok = v < nv
          slot = (it - p0) * threads + lanes
          assert slot.max() < slots * threads
          for s, vv in zip(slot[ok], v[ok]):
            assert s not in stage
            stage[int(s)] = int(vv)
        for it in range(p0, p1):
          v = it * span + first
          ok = v < nv
          slot = (it - p0) * threads + lanes
          for s, vv in zip(slot[ok], v[ok]):
            assert stage[int(s)] == int(vv)
            written[vv] += 1
        p0 += slots
    assert (written == 1).all()

  for nv, grid, threads, slots in ((1, 4, 8, 3), (95, 4, 8, 3), (96, 4, 8, 3), (97, 4, 8, 3), (1000, 4, 8, 3),
                                   (513, 8, 16, 10), (5000, 3, 32, 9), (32 * 7 * 10, 7, 32, 10)):
    replay(nv, grid, threads, slots)


def test_block_walk_covers_every_column_group_once():
  """bulyan_pass2_kernel and selected_mean_kernel (the second pass of a two-pass rule) walk blocks of T column groups in a
  grid-stride loop: v = b * T + lane, guarded by v < nv.  Replay: every group exactly once, every block T-aligned (whole
  cache lines per wave)."""
  for nv, grid, threads in ((1, 4, 8), (7, 1, 8), (8, 3, 8), (9, 3, 8), (1000, 7, 16), (4096, 16, 256), (4097, 5, 256)):
    nblk = (nv + threads - 1) // threads
    written = np.zeros(nv, dtype=np.int32)
    for block in range(grid):
      b = block
      while b < nblk:
        base = b * threads
        v = base + np.arange(threads)
        written[v[v < nv]] += 1
        b += grid
    assert (written == 1).all(), (nv, grid, threads)


def test_alloc_rows_layout_on_cpu():
  """layout.alloc_rows: rows of one allocation, 256-byte aligned starts, a pitch of (row bytes rounded up) + skew,
  successive calls continuing the skew sequence; short rows do not get a 2 MB pitch; bad arguments refused."""
  import torch
  from byzantinemomentum_amd import layout
  d = 300000  # 1.2 MB per row: the 2 MB pitch applies
  rows = layout.alloc_rows(5, d, "cpu", zero=True)
  assert len(rows) == 5 and all(r.shape == (d,) and r.is_contiguous() and r.dtype == torch.float32 for r in rows)
  assert float(sum(r.abs().sum() for r in rows)) == 0.0
  pitch = rows[1].data_ptr() - rows[0].data_ptr()
  assert pitch == (2 << 20) + layout.ROW_SKEW_BYTES and all(
    rows[i + 1].data_ptr() - rows[i].data_ptr() == pitch for i in range(4))
  assert rows[0].untyped_storage().data_ptr() == rows[4].untyped_storage().data_ptr()
  rows[2].fill_(1.0)  # rows do not overlap
  assert float(rows[1].sum()) == 0.0 and float(rows[3].sum()) == 0.0 and float(rows[2].sum()) == d
  more = layout.alloc_rows(3, d, "cpu")
  assert (more[0].data_ptr() - rows[0].data_ptr()) % 256 == 0
  short = layout.alloc_rows(4, 100, "cpu")
  assert short[1].data_ptr() - short[0].data_ptr() == 512 + layout.ROW_SKEW_BYTES  # 400 B rounded up to 256, plus the skew
  assert layout.alloc_rows(2, 0, "cpu")[1].shape == (0,)
  with pytest.raises(ValueError):
    layout.alloc_rows(0, 10, "cpu")
  with pytest.raises(ValueError):
    layout.alloc_rows(2, 10, "cpu", skew=100)


def test_graphed_call_needs_a_gpu():
  import byzantinemomentum_amd as bm
  if torch.cuda.is_available():
    pytest.skip("GPU present: covered by tests/test_gpu_parity_r3.py")
  with pytest.raises(RuntimeError, match="HIP graphs"):
    bm.graphs.GraphedCall(lambda: None)


# ---------------------------------------------------------------------------- #
# The device form of the Brute search (csrc/brute.hip), step by step in Python

def _device_brute_emulation(sq, n, f):
  """The control flow of brute_select_kernel with Python integers for the 64-bit row sets: threshold graphs,
  the search tree with its reduction rule (rows with more non-neighbours than removals left all go at once) and its
  branching rule (most non-neighbours, lowest index), the quickselect-style bisection (pivot = the middle open candidate
  of the middle row that still has one), the lexicographically first subset.  Returns (selection or None, probes)."""
  k = n - f
  fin = lambda v: abs(v) < math.inf  # noqa: E731
  dist = [[0.0] * n for _ in range(n)]
  for i in range(n):
    for j in range(n):
      if i != j:
        v = sq[min(i, j)][max(i, j)]
        dist[i][j] = math.sqrt(v) if (v == v and v != math.inf and v >= 0) else (v if v == v else math.nan)

  def build(t):
    return [sum(1 << j for j in range(n) if j != i and fin(dist[i][j]) and dist[i][j] <= t) for i in range(n)] + [0] * (64 - n)

  def has_clique(adj, cand, need):
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
      missing = [bin(cand & ~adj[l] & ~(1 << l)).count("1") if cand >> l & 1 else 0 for l in range(64)]
      if not any(m > 0 for m in missing):
        return True
      budget = count - need
      if budget == 0:
        continue
      forced = sum(1 << l for l in range(64) if missing[l] > budget)
      if forced:
        cand &= ~forced
        have = True
        continue
      worst = max(range(64), key=lambda l: (missing[l], -l))
      if missing[worst] <= budget:
        stack.append(cand & ~(1 << worst))
        cand &= adj[worst] | (1 << worst)
      else:
        cand &= ~(1 << worst)
      have = True

  everyone = (1 << n) - 1
  vmax = max([dist[i][j] for i in range(n) for j in range(i + 1, n) if fin(dist[i][j])] + [0.0])
  if not has_clique(build(vmax), everyone, k):
    return None, 0
  lo, hi, probes = -1.0, vmax, 0
  if has_clique(build(0.0), everyone, k):
    hi = 0.0
  else:
    lo = 0.0
    while True:
      per = [[dist[i][j] for j in range(i + 1, n) if fin(dist[i][j]) and lo < dist[i][j] < hi] for i in range(n)]
      holders = [i for i in range(n) if per[i]]
      if not holders:
        break
      src = holders[len(holders) // 2]
      pivot = per[src][len(per[src]) // 2]
      probes += 1
      if has_clique(build(pivot), everyone, k):
        hi = pivot
      else:
        lo = pivot
  adj = build(hi)
  cand, sel = everyone, []
  for c in range(n):
    if len(sel) >= k:
      break
    if not cand >> c & 1:
      continue
    nxt = cand & adj[c] & ~((1 << (c + 1)) - 1)
    if has_clique(adj, nxt, k - len(sel) - 1):
      sel.append(c)
      cand = nxt
  return (sel if len(sel) == k else None), probes


def test_device_brute_algorithm_equals_the_host_search():
  """The algorithm of the device kernel (emulated) against bm_brute_select — itself pinned on exhaustive enumeration
  above — on random matrices, matrices of few distinct values, clusters with aliased rows, rows at non-finite distance
  and the case without any finite subset.  (The kernel itself is compared with the host search on the same 88
  matrices on the GPU, tests/test_gpu_parity_r4.py.)"""
  from byzantinemomentum_amd import gars
  gen = torch.Generator().manual_seed(7)
  worst = 0
  for n, f in ((4, 1), (7, 2), (11, 2), (11, 4), (25, 5), (25, 11), (33, 8), (51, 12), (64, 20), (64, 1), (9, 0)):
    for variant in range(8):
      if variant % 4 == 0:
        pts = torch.randn(n, 6, generator=gen, dtype=torch.float64)
      elif variant % 4 == 1:
        pts = torch.randint(0, 3, (n, 4), generator=gen).double()
      elif variant % 4 == 2:
        pts = torch.randn(n, 5, generator=gen, dtype=torch.float64)
        pts[: n - f] *= 0.01
        pts[-1] = pts[-2]
      else:
        pts = torch.rand(n, 3, generator=gen, dtype=torch.float64).round(decimals=1)
      sq = (pts[:, None, :] - pts[None, :, :]).pow(2).sum(dim=2)
      if variant >= 4 and f >= 1:
        for r in range(min(f, 2) if variant < 6 else f + 1):
          row = (3 * r + 1) % n
          sq[row, :] = math.nan if r % 2 == 0 else math.inf
          sq[:, row] = sq[row, :]
      try:
        want = gars.brute_select_host(sq.sqrt().contiguous(), n, f)
      except RuntimeError:
        want = None
      got, probes = _device_brute_emulation(sq.tolist(), n, f)
      assert got == want, (n, f, variant, got, want)
      worst = max(worst, probes)
  assert worst <= 24  # the bisection stays logarithmic on these inputs (2 016 candidates at n = 64)


# ---------------------------------------------------------------------------- #
# The device ranking (csrc/rank_body.h), step by step in Python

def _device_rank_emulation(sq, n, f, m, krum_mode):
  """krum_rank_body with Python floats: per row, the 64 lanes hold sqrt(sq) (non-finite -> +inf, own lane and lanes
  past n +inf), the bitonic network of the kernel (same k / j loops, same keep-min predicate) sorts them across the
  lanes, lane i adds the `take` smallest of row i in ascending order, rows are ranked by (score, index)."""
  inf = math.inf
  srt = []
  for i in range(n):
    v = [inf] * 64
    for lane in range(n):
      if lane != i:
        s = sq[i][lane]
        r = math.sqrt(s) if s >= 0 else math.nan  # (sqrt of NaN / negative: NaN, as the device's sqrt)
        v[lane] = inf if (r != r or abs(r) == inf) else r
    k = 2
    while k <= 64:
      j = k >> 1
      while j > 0:
        o = [v[lane ^ j] for lane in range(64)]
        keep_min = [((lane & k) == 0) == ((lane & j) == 0) for lane in range(64)]
        v = [min(v[lane], o[lane]) if keep_min[lane] else max(v[lane], o[lane]) for lane in range(64)]
        j >>= 1
      k <<= 1
    assert all(v[t] <= v[t + 1] for t in range(63)), "the network does not sort"
    srt.append(v[:n - 1])
  take = max(0, min(n - 1, (n - f - 1) if krum_mode else m))
  scores = []
  for i in range(n):
    s = 0.0
    for t in range(take):
      s += srt[i][t]
    scores.append(s)
  order = [None] * n
  for i in range(n):
    rank = sum(1 for j in range(n) if scores[j] < scores[i] or (scores[j] == scores[i] and j < i))
    order[rank] = i
  return order, scores


def test_device_rank_algorithm_equals_the_oracle():
  """The algorithm of krum_rank_body (emulated: bitonic network across 64 lanes, ascending fp64 sums, rank by counting)
  against the oracle's restatement of krum.py:50-62 / bulyan.py:56-62 on random matrices, matrices with exact ties
  (aliased rows, few distinct values), rows at non-finite distance, and the smallest stacks.  (The kernel itself is
  compared with the oracle on the GPU, tests/test_gpu_parity*.py.)"""
  gen = torch.Generator().manual_seed(11)
  for n, f in ((1, 0), (2, 0), (3, 0), (5, 1), (7, 2), (11, 2), (11, 4), (25, 5), (25, 11), (33, 8), (51, 12), (64, 20), (64, 30)):
    for variant in range(6):
      if variant % 3 == 0:
        pts = torch.randn(n, 6, generator=gen, dtype=torch.float64)
      elif variant % 3 == 1:
        pts = torch.randint(0, 3, (n, 3), generator=gen).double()  # few distinct distances: exact score ties
      else:
        pts = torch.randn(n, 5, generator=gen, dtype=torch.float64)
        if n >= 3:
          pts[-1] = pts[-2]                                           # aliased rows: zero distance, tied scores
      sq = (pts[:, None, :] - pts[None, :, :]).pow(2).sum(dim=2)
      if variant >= 3 and n >= 3:
        row = (3 * variant + 1) % n
        sq[row, :] = math.nan if variant % 2 else math.inf
        sq[:, row] = sq[row, :]
      # (math.sqrt like the emulation: torch's vectorised fp64 sqrt is not correctly rounded on every CPU path, and the
      #  subject here is the ranking, not the square root)
      dist = np.array([[math.sqrt(v) if v >= 0 else math.nan for v in row] for row in sq.tolist()])
      dist[~np.isfinite(dist)] = math.inf                             # krum.py:46-47
      np.fill_diagonal(dist, 0.0)
      m = max(1, n - f - 2)
      # Krum scores: n-f-1 smallest of the n-1 others (krum.py:59-60); Bulyan: m smallest (bulyan.py:58-61)
      for krum_mode in (True, False):
        got_order, got_scores = _device_rank_emulation(sq.tolist(), n, f, m, krum_mode)
        if krum_mode:
          want_scores = O.krum_scores(dist, f) if n - f - 1 >= 0 else None
        else:
          want_scores = [O._sum_smallest([dist[i, j] if j != i else math.inf for j in range(n)], min(m, n - 1)) for i in range(n)]
        if want_scores is None:
          continue
        assert got_scores == list(want_scores), (n, f, variant, krum_mode)
        assert got_order == O._stable_order(list(want_scores)), (n, f, variant, krum_mode)


def test_brute_status_codes_raise_what_they_mean():
  """The device search's status: 0 passes, -1 is the reference's failed assertion (brute.py:68), -2 the node budget."""
  import torch
  from byzantinemomentum_amd import gars
  gars.brute_check(torch.tensor([0], dtype=torch.int32))
  gars.brute_check(None if gars.last_brute_status is None else torch.tensor([0], dtype=torch.int32))
  with pytest.raises(RuntimeError, match="no subset of n-f rows has a finite diameter"):
    gars.brute_check(torch.tensor([-1], dtype=torch.int32))
  with pytest.raises(RuntimeError, match="budget"):
    gars.brute_check(torch.tensor([-2], dtype=torch.int32))
