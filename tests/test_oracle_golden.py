"""The oracle (f32 mode) must reproduce, bit for bit, what the real reference returned when the
fixtures were generated.  Runs everywhere (no reference checkout, no GPU needed)."""

import math

import numpy as np
import pytest

from oracle import gar_oracle as O
from tests.golden_io import CASES, HAND_CASES, Golden, same_bits


@pytest.mark.parametrize("name", CASES + HAND_CASES)
def test_colwise(name):
  g = Golden(name)
  assert same_bits(O.median(g.gradients), g.tensor("median"))
  for rule in ("trmean", "phocas", "meamed"):
    if g.has(rule):
      assert same_bits(getattr(O, rule)(g.gradients, g.f), g.tensor(rule)), rule


@pytest.mark.parametrize("name", CASES)
def test_krum_bulyan(name):
  g = Golden(name)
  order, scores = O.krum_order(g.gradients, g.f)
  assert order == g.array("krum_order").tolist()
  assert [scores[i] for i in order] == g.array("krum_scores").tolist()
  assert same_bits(O.krum(g.gradients, g.f), g.tensor("krum"))
  assert same_bits(O.krum(g.gradients, g.f, 1), g.tensor("krum_m1"))
  if g.has("bulyan"):
    assert same_bits(O.bulyan(g.gradients, g.f), g.tensor("bulyan"))
    assert same_bits(O.bulyan(g.gradients, g.f, 3), g.tensor("bulyan_m3"))


@pytest.mark.parametrize("name", CASES)
def test_brute_aksel_average_cge(name):
  g = Golden(name)
  if g.has("brute"):
    assert O.brute_selection(g.gradients, g.f) == g.array("brute_selection").tolist()
    assert same_bits(O.brute(g.gradients, g.f), g.tensor("brute"))
  order, sq = O.aksel_order(g.gradients)
  assert order == g.array("aksel_order").tolist()
  want = g.array("aksel_sqdist")
  got = np.array([sq[i] for i in order])
  assert np.array_equal(got, want, equal_nan=True)
  assert same_bits(O.aksel(g.gradients, g.f, "mid"), g.tensor("aksel_mid"))
  assert same_bits(O.aksel(g.gradients, g.f, "n-f"), g.tensor("aksel_n-f"))
  assert same_bits(O.average(g.gradients), g.tensor("average"))
  assert same_bits(O.cge(g.gradients, g.f), g.tensor("cge"))


@pytest.mark.parametrize("name", CASES)
def test_stats(name):
  g = Golden(name)
  for prefix, samples in (("honest", g.honests), ("attack", g.attacks)):
    if not g.has(prefix + "_stats"):
      continue
    avg, norm, dev, mx = O.compute_avg_dev_max(samples)
    assert same_bits(avg, g.tensor(prefix + "_avg"))
    for a, b in zip((norm, dev, mx), g.array(prefix + "_stats").tolist()):
      assert (math.isnan(a) and math.isnan(b)) or a == b


def test_closest_window_matches_topk_formulation():
  """The window formulation used by the kernels equals the reference's topk formulation wherever the
  choice is not ambiguous (no exact tie at the window edge)."""
  import torch
  g = Golden("little_n25_f5")
  st = torch.stack(g.gradients)
  centre = st.median(dim=0).values
  ref = O.meamed(g.gradients, g.f).to(torch.float64)
  win, amb = O.closest_window(st, g.n - g.f, centre)
  ok = (win - ref).abs() <= 1e-6 * (1 + ref.abs())
  assert bool((ok | amb).all())
  assert int(amb.sum()) < g.d // 10


# ---------------------------------------------------------------------------- #
# Seeded `study` files of the unmodified reference driver (golden vector 2)

def _study_rows(text):
  lines = [ln for ln in text.splitlines() if ln.strip()]
  assert lines[0].startswith("# Step number") and len(lines[0].split("\t")) == 24
  return [ln.split("\t") for ln in lines[1:]]


def test_study_goldens_are_well_formed():
  """The committed study files (scripts/make_golden_study.py): 24 columns, 4 steps, NaN exactly where attack.py
  prints NaN on the first step (no previous sampled average yet: columns 21, 22)."""
  import glob
  import math
  import os
  files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "study_*.tsv")))
  assert len(files) == 3
  for path in files:
    rows = _study_rows(open(path).read())
    assert len(rows) == 4 and all(len(r) == 24 for r in rows)
    assert [int(r[0]) for r in rows] == [0, 1, 2, 3]
    assert math.isnan(float(rows[0][21])) and math.isnan(float(rows[0][22]))
    assert all(math.isfinite(float(x)) for r in rows[1:] for x in r[2:23])
    assert float(rows[0][3]) == 0.0  # l2 from origin before the first update


@pytest.mark.reference
def test_study_golden_reproduces():
  """Where the reference checkout is present: the same seeded commands give the committed files again (to 1e-6: the CPU
  matmuls of the model may add in another order on another core count)."""
  import importlib.util
  import math
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location("make_golden_study", os.path.join(root, "scripts", "make_golden_study.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  name = "krum_n11_f2_empire_worker"
  got = _study_rows(mod.run_case(mod.CASES[name]))
  want = _study_rows(open(os.path.join(root, "tests", "golden", f"study_{name}.tsv")).read())
  assert len(got) == len(want)
  for rg, rw in zip(got, want):
    assert rg[:2] == rw[:2] and rg[23] == rw[23]  # step, points, accepted ratio
    for a, b in zip(rg[2:23], rw[2:23]):
      a, b = float(a), float(b)
      assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= 1e-6 * max(abs(b), 1e-3), (rg[0], a, b)
