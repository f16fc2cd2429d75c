"""Pin the oracle: in "f32" mode it must be BIT-IDENTICAL to the real reference, imported
unmodified from /root/reference (only possible in the build container; the committed fixtures of
tests/golden/ carry the same pin to the GPU box, see test_oracle_golden.py)."""

import math
import sys

import pytest
import torch

from oracle import gar_oracle as O
from oracle import reference_loader

pytestmark = pytest.mark.reference

SHAPES = [("hetero", 11, 2, 1000, 1), ("hetero", 25, 5, 777, 2), ("little", 25, 5, 500, 3),
          ("iid", 13, 2, 300, 4), ("hetero", 51, 12, 200, 5), ("nan", 11, 2, 100, 6)]


@pytest.fixture(scope="module")
def ref():
  return reference_loader.load(with_native=False)[0]


@pytest.mark.parametrize("kind,n,f,d,seed", SHAPES)
def test_colwise_rules(ref, kind, n, f, d, seed):
  g, _ = O.make_stack(kind, n, f, d, seed)
  for name, fn in (("median", O.median), ("trmean", O.trmean), ("phocas", O.phocas), ("meamed", O.meamed)):
    kw = {} if name == "median" else {"f": f}
    want = ref.gars[name].unchecked(gradients=g, **kw)
    got = fn(g, **kw)
    assert torch.equal(torch.nan_to_num(got, nan=1234.5), torch.nan_to_num(want, nan=1234.5)), name


@pytest.mark.parametrize("kind,n,f,d,seed", SHAPES)
def test_krum(ref, kind, n, f, d, seed):
  g, _ = O.make_stack(kind, n, f, d, seed)
  for m in (None, 1, max(1, (n - f - 2) // 2)):
    want = ref.gars["krum"].unchecked(gradients=g, f=f, m=m)
    got = O.krum(g, f, m)
    assert torch.equal(got, want)
  # score order is what `influence` is made of
  rk = sys.modules["aggregators.krum"]  # the package attribute `krum` is the GAR function itself
  scores = rk._compute_scores(g, f, None)
  order, vals = O.krum_order(g, f)
  assert [s for s, _ in scores] == [vals[i] for i in order]
  assert all(gr is g[i] for (_, gr), i in zip(scores, order))


@pytest.mark.parametrize("kind,n,f,d,seed", [s for s in SHAPES if s[1] >= 4 * s[2] + 3])
def test_bulyan(ref, kind, n, f, d, seed):
  g, _ = O.make_stack(kind, n, f, d, seed)
  for m in (None, 1, 3, n - f - 2):
    want = ref.gars["bulyan"].unchecked(gradients=g, f=f, m=m)
    got = O.bulyan(g, f, m)
    assert torch.equal(torch.nan_to_num(got, nan=1234.5), torch.nan_to_num(want, nan=1234.5)), m


@pytest.mark.parametrize("kind,n,f,d,seed", [("hetero", 11, 2, 300, 7), ("little", 9, 3, 100, 8), ("nan", 9, 2, 50, 9)])
def test_brute(ref, kind, n, f, d, seed):
  g, _ = O.make_stack(kind, n, f, d, seed)
  rb = sys.modules["aggregators.brute"]
  assert list(rb._compute_selection(g, f)) == O.brute_selection(g, f)
  assert torch.equal(O.brute(g, f), ref.gars["brute"].unchecked(gradients=g, f=f))


@pytest.mark.parametrize("kind,n,f,d,seed", SHAPES[:5])
def test_aksel_average_cge(ref, kind, n, f, d, seed):
  g, _ = O.make_stack(kind, n, f, d, seed)
  for mode in ("mid", "n-f"):
    assert torch.equal(O.aksel(g, f, mode), ref.gars["aksel"].unchecked(gradients=g, f=f, mode=mode))
  assert torch.equal(O.average(g), ref.gars["average"].unchecked(gradients=g))
  assert torch.equal(O.cge(g, f), ref.gars["cge"].unchecked(gradients=g, f=f))


def test_compute_avg_dev_max():
  tools = reference_loader.load(with_native=False)[1]
  for kind, n, f, d, seed in SHAPES[:4]:
    g, h = O.make_stack(kind, n, f, d, seed)
    for samples in (g[:h], g[h:], g[:1], []):
      want = tools.compute_avg_dev_max(samples)
      got = O.compute_avg_dev_max(samples)
      if want[0] is None:
        assert got[0] is None
      else:
        assert torch.equal(got[0], want[0])
      for a, b in zip(got[1:], want[1:]):
        assert (math.isnan(a) and math.isnan(b)) or a == b


def test_f64_mode_agrees_on_separated_inputs():
  g, _ = O.make_stack("hetero", 25, 5, 4000, 11)
  assert O.krum_order(g, 5, "f32")[0] == O.krum_order(g, 5, "f64")[0]
  assert O.bulyan_order(g, 5, None, "f32")[0] == O.bulyan_order(g, 5, None, "f64")[0]
  assert O.aksel_order(g, "f32")[0] == O.aksel_order(g, "f64")[0]
