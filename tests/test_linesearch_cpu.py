"""The factor search of the "identical" attacks on the host (csrc/linesearch.cpp, no GPU involved):
the bm_search_* cursor against the restatement of tools.line_maximize, and the scalar form of the search
(bm_attack_objective / bm_attack_line_search) against the search run the reference's way — the rule
evaluated on the vectors once per candidate — on float64 distances."""

import ctypes
import math
import random

import pytest
import torch

from byzantinemomentum_amd import _lib, linesearch
from oracle import gar_oracle as O


def scapes(seed):
  rng = random.Random(seed)
  a, b, c = (rng.uniform(0, 30) for _ in range(3))
  return [lambda x: -(x - b) ** 2, lambda x: math.sin(x) + 0.1 * x, lambda x: min(x, a),  # the last ones: plateaus
          lambda x: -x, lambda x: 1.0, lambda x: math.nan if x > c else x]


@pytest.mark.parametrize("evals", [1, 2, 3, 7, 16, 40])
def test_line_maximize_equals_restatement(evals):
  for seed in range(40):
    for scape in scapes(seed):
      got, got_trace = linesearch.line_maximize(scape, evals=evals)
      want, want_trace = O.line_maximize(scape, evals=evals)
      assert got == want
      assert len(got_trace) == evals and all(
        x == xo and (y == yo or (math.isnan(y) and math.isnan(yo))) for (x, y), (xo, yo) in zip(got_trace, want_trace))


def test_line_maximize_other_parameters_and_errors():
  f = lambda x: -(x - 11.3) ** 2  # noqa: E731
  for start, delta, ratio in ((0.0, 1.0, 0.8), (2.5, 0.25, 0.6), (20.0, 4.0, 0.95)):
    assert linesearch.line_maximize(f, 12, start, delta, ratio)[0] == O.line_maximize(f, 12, start, delta, ratio)[0]
  for bad in (dict(evals=0), dict(start=-1.0), dict(delta=0.0), dict(ratio=0.5), dict(ratio=1.0)):
    with pytest.raises(RuntimeError, match="invalid argument"):
      linesearch.line_maximize(f, **bad)

  def boom(x):
    raise KeyError("from the callback")
  with pytest.raises(KeyError):  # an exception of the callable propagates (the evaluations are the caller's)
    linesearch.line_maximize(boom, evals=4)


def test_search_cursor_protocol():
  """One report per proposal, in that order; the cursor is plain caller-owned data."""
  import ctypes
  lib = _lib.load()
  cur = _lib.Search()
  x = ctypes.c_double()
  assert lib.bm_search_begin(ctypes.byref(cur), 0.0, 1.0, 0.8) == 0
  assert lib.bm_search_report(ctypes.byref(cur), 1.0) == _lib.EINVAL       # nothing proposed yet
  assert lib.bm_search_propose(ctypes.byref(cur), ctypes.byref(x)) == 0 and x.value == 0.0
  assert lib.bm_search_propose(ctypes.byref(cur), ctypes.byref(x)) == _lib.EINVAL  # the first one is unanswered
  assert lib.bm_search_report(ctypes.byref(cur), 3.0) == 0
  assert (cur.evaluations, cur.best_x, cur.best_y) == (1, 0.0, 3.0)
  assert lib.bm_search_propose(ctypes.byref(cur), ctypes.byref(x)) == 0 and x.value == 1.0
  assert lib.bm_search_begin(None, 0.0, 1.0, 0.8) == _lib.EINVAL


def honest_stack(seed, h, d=400):
  gen = torch.Generator().manual_seed(seed)
  base = 0.3 * torch.randn(d, generator=gen)
  return [base + (0.5 + 0.1 * i) * torch.randn(d, generator=gen) for i in range(h)]


def ext_matrix(honests, kind):
  """float64 squared distances among the honest rows, their mean and mean + direction (what the product
  obtains from one bm_pairwise_sqdist over h+2 rows)."""
  stck = torch.stack(honests)
  avg = stck.mean(dim=0)
  att = avg.neg() if kind == "empire" else stck.var(dim=0).sqrt_()
  rows = torch.stack([g.double() for g in honests] + [avg.double(), (avg + att).double()])
  return ((rows[:, None, :] - rows[None, :, :]) ** 2).sum(dim=2).contiguous()


RULES = {"krum": lambda g, f: O.krum(g, f), "brute": lambda g, f: O.brute(g, f), "average": lambda g, f: O.average(g)}


@pytest.mark.parametrize("rule", ["krum", "brute", "average"])
@pytest.mark.parametrize("kind", ["empire", "little"])
def test_scalar_search_equals_vector_search(rule, kind):
  ties = 0
  for seed in range(12):
    n, f = ((7, 1), (9, 2), (11, 2), (15, 3))[seed % 4]
    if rule == "brute" and n > 11:
      n, f = 11, 2
    h = n - f
    honests = honest_stack(seed, h)
    ext = ext_matrix(honests, kind)
    for negative in (False, True):
      factor, trace = linesearch.attack_line_search(ext, h, f, f, rule, evals=16, negative=negative)
      _, want, want_trace = O.identical_attack(honests, f, f, RULES[rule], kind, -16, negative, "f64")
      floor = 1e-9 * max(y for _, y in want_trace)
      if factor != want:  # only legitimate on a tie of the two best candidates
        ties += 1
        assert abs(max(y for _, y in trace) - max(y for _, y in want_trace)) <= 1e-5 * max(y for _, y in want_trace)
        continue
      for (x, y), (xo, yo) in zip(trace, want_trace):
        assert x == xo and abs(y - yo) <= 2e-5 * abs(yo) + floor, (seed, negative, x, y, yo)
  assert ties <= 2


def test_objective_reports_the_selection():
  n, f = 11, 2
  h = n - f
  honests = honest_stack(3, h)
  ext = ext_matrix(honests, "empire")
  for t in (0.0, 0.7, 1.0, 3.0, 40.0):
    stck = torch.stack(honests)
    avg = stck.mean(dim=0)
    cand = avg + t * avg.neg()
    grads = honests + [cand] * f
    order, _ = O.krum_order(grads, f, "f64")
    y, sel = linesearch.attack_objective(ext, h, f, f, "krum", t)
    assert sel == order[:n - f - 2]
    want = (O.krum(grads, f, precision="f64") - avg.double()).pow(2).sum().item()
    assert abs(y - want) <= 1e-6 * want + 1e-12  # the candidate vector is rounded to fp32, the scalar form is not
    y1, sel1 = linesearch.attack_objective(ext, h, f, f, "krum", t, m=1)
    assert sel1 == order[:1]
    _, sel_b = linesearch.attack_objective(ext, h, f, f, "brute", t)
    assert sel_b == list(O.brute_selection(grads, f, "f64"))
    _, sel_a = linesearch.attack_objective(ext, h, f, f, "average", t)
    assert sel_a == list(range(n))


def test_scalar_search_rejects_bad_arguments():
  honests = honest_stack(0, 5)
  ext = ext_matrix(honests, "empire")
  with pytest.raises(ValueError):
    linesearch.attack_line_search(ext, 5, 1, 1, "median")
  with pytest.raises(ValueError):
    linesearch.attack_line_search(ext.float(), 5, 1, 1, "krum")
  with pytest.raises(ValueError):
    linesearch.attack_line_search(ext, 6, 1, 1, "krum")  # shape does not match h
  lib = _lib.load()
  y = (torch.zeros(1, dtype=torch.float64))
  for args in ((0, 1, 1, 0, 0), (5, -1, 1, 0, 0), (63, 2, 1, 0, 0), (5, 1, 1, 2, 0), (5, 1, 1, 0, 7), (5, 1, 7, 6, 0)):
    h, k, f, rule, m = args
    rc = lib.bm_attack_objective(ext.data_ptr(), h, k, f, rule, m, 1.0, y.data_ptr(), None, None)
    assert rc == _lib.EINVAL, args


@pytest.mark.parametrize("kind", ["empire", "little"])
def test_ranking_from_scalars_equals_ranking_of_the_vectors(kind):
  """bm_attack_ranking: the order bm_krum_rank gives for honests + [avg + t*att] * k, computed from the (h+2)^2 scalars —
  against the oracle's Krum / Bulyan ranking (krum.py:41-62, bulyan.py:48-62) of the actual candidate stack in fp64."""
  for seed in range(8):
    n, f = ((11, 2), (15, 3), (25, 5), (9, 1))[seed % 4]
    h = n - f
    honests = honest_stack(seed, h)
    ext = ext_matrix(honests, kind)
    stck = torch.stack(honests)
    avg = stck.mean(dim=0)
    att = avg.neg() if kind == "empire" else stck.var(dim=0).sqrt_()
    for t in (0.0, 0.3, 1.1, -2.0, 25.0):
      grads = honests + [avg + t * att] * f
      want_k, _ = O.krum_order(grads, f, "f64")
      assert linesearch.attack_ranking(ext, h, f, f, "krum", t) == list(want_k)
      for m in (None, 3):
        want_b, _ = O.bulyan_order(grads, f, m, "f64")
        assert linesearch.attack_ranking(ext, h, f, f, "bulyan", t, m) == list(want_b), (seed, t, m)
  lib = _lib.load()
  buf = (ctypes.c_int32 * 64)()
  honests = honest_stack(0, 5)
  ext = ext_matrix(honests, "empire")
  for args in ((0, 1, 1, 0, 0), (5, -1, 1, 0, 0), (63, 2, 1, 0, 0), (5, 1, 1, 2, 0), (5, 1, 1, 1, 7)):
    h, k, f, mode, m = args
    assert lib.bm_attack_ranking(ext.data_ptr(), h, k, f, mode, m, 1.0, ctypes.cast(buf, ctypes.c_void_p)) == _lib.EINVAL
