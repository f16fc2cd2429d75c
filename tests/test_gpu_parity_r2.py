"""Round-2 GPU parity: the BASELINE.json configurations at FULL size against fp64 computed with torch
on the same GPU, ill-conditioned stacks in every BM_PAIR_MODE, the closest-to-centre rules without
exempt columns, and the plugin objects exactly as the reference's call sites use them.
Needs an MI355X: `pytest -m gpu`.
"""

import math
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from oracle import gar_oracle as O
from tests.golden_io import CASES, Golden, same_bits

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D_RESNET18 = 11173962
D_WRN = 36546980


@pytest.fixture(scope="module")
def bm():
  import byzantinemomentum_amd
  byzantinemomentum_amd._lib.load()
  return byzantinemomentum_amd


def to_dev(gradients):
  seen = {}
  return [seen.setdefault(id(g), g.to(DEV)) for g in gradients]


def gpu_stack(kind, n, f, d, seed):
  """oracle.make_stack's "hetero"/"tight" distributions generated on the GPU (full-size cases)."""
  gen = torch.Generator(device=DEV).manual_seed(seed)
  h = n - f
  if kind == "tight":
    mu = 10.0 * torch.randn(d, device=DEV, generator=gen)
    sig = torch.linspace(0.01, 0.1, h).tolist()
  else:
    mu = 0.1 * torch.randn(d, device=DEV, generator=gen)
    sig = torch.linspace(0.5, 1.5, h).tolist()
  honest = [mu + s * torch.randn(d, device=DEV, generator=gen) for s in sig]
  acc = torch.zeros(d, dtype=torch.float32, device=DEV)
  for g in honest:
    acc += g
  byz = acc.div_(h).mul_(-0.1)
  return honest + [byz] * f, h


def sqdist_f64_on_gpu(rows):
  """n x n float64 squared distances, direct differences in fp64 on the GPU (no Gram, no cancellation)."""
  n = len(rows)
  st = torch.stack([r.double() for r in rows])
  out = np.zeros((n, n))
  for i in range(n - 1):
    diff = st[i + 1:] - st[i]
    vals = (diff * diff).sum(dim=1).cpu().numpy()
    out[i, i + 1:] = vals
    out[i + 1:, i] = vals
    del diff
  return out


def decisive(scores, k, tol=1e-5):
  srt = sorted(scores)
  return k >= len(srt) or srt[k] - srt[k - 1] > tol * abs(srt[k])


# ---------------------------------------------------------------------------- #
# BASELINE config 3 at full size: n = 51, f = 12, d = 11 173 962

@pytest.mark.parametrize("kind", ["hetero", "tight"])
def test_full_size_c3_krum_against_fp64(bm, kind):
  n, f, d = 51, 12, D_RESNET18
  m = n - f - 2
  rows, h = gpu_stack(kind, n, f, d, seed=2024)
  want_sq = sqdist_f64_on_gpu(rows)                       # all 1 275 distances, fp64, same GPU
  sq = bm.gars.pairwise_sqdist(rows).cpu().numpy()
  off = ~np.eye(n, dtype=bool) & (want_sq > 0)
  rel = np.abs(sq - want_sq)[off] / want_sq[off]
  assert rel.max() <= 1e-5, f"{kind}: worst relative error of a squared distance {rel.max():.2e}"
  for a in range(h + 1, n):
    assert sq[h, a] == 0.0 and np.array_equal(sq[h, :h], sq[a, :h])
  # rank with the oracle's own logic (krum.py:50-62) on the fp64 distances
  scores = O.krum_scores(np.sqrt(want_sq), f)
  order = O._stable_order(scores)
  srt = sorted(scores)
  assert srt[m] - srt[m - 1] > 1e-5 * srt[m], "generator is meant to separate the selected set decisively"
  got = bm.gars.krum_selection(rows, f)
  assert sorted(got) == sorted(order[:m])
  # same order too, up to permutations inside runs of scores that agree to 1e-5 (the float64 reference sums
  # themselves carry ~1e-9 of rounding); exactly tied rows (the aliased Byzantine gradients) come in index order
  pos = {r: k for k, r in enumerate(order)}
  for k, r in enumerate(got):
    lo, hi = sorted((k, pos[r]))
    assert all(scores[order[t + 1]] - scores[order[t]] <= 1e-5 * scores[order[t + 1]] for t in range(lo, hi)), (k, r)
  byz = [r for r in got if r >= h]
  assert byz == sorted(byz)
  # the average: torch's own sequential fp32 sum on the same GPU in that order, true division on the host
  acc = torch.zeros(d, dtype=torch.float32, device=DEV)
  for i in got:
    acc = acc + rows[i]
  assert same_bits(bm.krum(rows, f), acc.cpu().div_(m))
  acc1 = (torch.zeros(d, dtype=torch.float32, device=DEV) + rows[got[0]]).cpu().div_(1)
  assert same_bits(bm.krum(rows, f, 1), acc1)


# ---------------------------------------------------------------------------- #
# BASELINE config 4 at full size on one GPU: Bulyan n = 25, f = 5

def test_full_size_c4_bulyan_against_fp64(bm):
  n, f, d = 25, 5, D_RESNET18
  m = n - f - 2
  theta, beta = n - 2 * f - 2, n - 4 * f - 2
  rows, h = gpu_stack("hetero", n, f, d, seed=4)
  want_sq = sqdist_f64_on_gpu(rows)
  # bulyan.py:56-62: score = sum of the m smallest distances of the row
  dist = np.sqrt(want_sq)
  scores = [O._sum_smallest([dist[i, j] for j in range(n) if j != i], m) for i in range(n)]
  order = O._stable_order(scores)
  got = bm.gars.bulyan_ranking(rows, f)
  # neighbours in the ranking are either exactly tied (aliased Byzantine rows: index order) or well apart
  assert all(scores[a] == scores[b] or scores[b] - scores[a] > 1e-5 * scores[b] for a, b in zip(order, order[1:]))
  assert got == order
  # pass 2 with the reference's own fp32 arithmetic on the GPU (bulyan.py:64-84, static scores): sequential
  # sums in rank order, true division (tensor / tensor: torch's scalar division multiplies by a reciprocal)
  sel = []
  for i in range(theta):
    cnt = min(m, m - i)
    acc = torch.zeros(d, dtype=torch.float32, device=DEV)
    for r in got[i:i + cnt]:
      acc = acc + rows[r]
    sel.append(torch.div(acc, torch.full_like(acc, float(cnt))))
  sel = torch.stack(sel)
  med = sel.median(dim=0).values
  dev = (sel - med).abs()
  srt = dev.sort(dim=0).values
  idx = dev.topk(beta, dim=0, largest=False, sorted=False).indices
  want = sel.gather(0, idx).double().mean(dim=0)
  out = bm.bulyan(rows, f)
  scale = float(torch.stack([r.abs().max() for r in rows[:h]]).max())
  bad = (out.double() - want).abs() > 2e-6 * scale
  # the only legitimate disagreement: the beta-th and (beta+1)-th deviations tie, topk may keep either
  tie = srt[beta - 1] == srt[beta]
  assert int((bad & ~tie).sum()) == 0, int((bad & ~tie).sum())
  assert int(tie.sum()) <= d // 1000, ("tie columns", int(tie.sum()))


# ---------------------------------------------------------------------------- #
# Ill-conditioned stacks in every distance mode (the library reads BM_PAIR_MODE once per process)

@pytest.mark.parametrize("mode", ["0", "1"])
def test_tight_and_momentum_stacks_in_every_pair_mode(mode):
  env = dict(os.environ, BM_PAIR_MODE=mode, PYTHONPATH=ROOT)
  out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pair_mode_check.py")], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=900)
  assert out.returncode == 0 and "pair-mode ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
  print(out.stdout.strip())


def test_accuracy_gate_hands_over_to_the_direct_kernel():
  """BM_PAIR_TAU=1e30 lists every row: the gated direct kernel must then produce what BM_PAIR_MODE=1
  produces (same kernel; the grid may differ, hence equality to fp64 rounding, not bitwise)."""
  code = (
    "import torch, sys\n"
    "import byzantinemomentum_amd as bm\n"
    "from oracle import gar_oracle as O\n"
    "rows, h = O.make_stack('hetero', 25, 5, 70001, seed=8)\n"
    "seen = {}\n"
    "dev = [seen.setdefault(id(g), g.to('cuda:0')) for g in rows]\n"
    "torch.save(bm.gars.pairwise_sqdist(dev).cpu(), sys.argv[1])\n")
  import tempfile
  with tempfile.TemporaryDirectory() as tmp:
    paths = []
    for name, extra in (("gated", {"BM_PAIR_TAU": "1e30"}), ("direct", {"BM_PAIR_MODE": "1"})):
      path = os.path.join(tmp, name + ".pt")
      env = dict(os.environ, PYTHONPATH=ROOT, **extra)
      out = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=env, cwd=ROOT)
      assert out.returncode == 0, out.stderr[-2000:]
      paths.append(path)
    a, b = torch.load(paths[0]), torch.load(paths[1])
    assert float(((a - b).abs() / b.clamp(min=1e-300)).max()) <= 1e-9 and torch.equal(a, a.T)


# ---------------------------------------------------------------------------- #
# phocas / meamed: no exempt columns — an exact deviation tie must give one of the two legal windows

def _window_candidates(st, keep, centre):
  """(mean of the reference-legal window, ambiguous mask, list of alternative means) per column."""
  g = st.to(torch.float64)
  c = centre.to(torch.float64)
  n, d = g.shape
  srt = g.sort(dim=0).values
  win, amb = O.closest_window(st, keep, centre)
  # alternatives: every contiguous window of `keep` sorted values (the topk result is always one of them
  # when deviations tie only at the window edges)
  csum = torch.cat([torch.zeros(1, d, dtype=torch.float64), srt.cumsum(dim=0)])
  alts = [(csum[s + keep] - csum[s]) / keep for s in range(n - keep + 1)]
  return win, amb, alts


@pytest.mark.parametrize("name", CASES)
def test_closest_rules_without_exemptions(bm, name):
  g = Golden(name)
  if not g.has("trmean"):
    pytest.skip("no trimmed-mean fixture")
  dev = to_dev(g.gradients)
  st = torch.stack(g.gradients)
  finite = torch.isfinite(st)
  scale = float(st[finite].abs().max())
  for rule, centre in (("phocas", O.trmean(g.gradients, g.f)), ("meamed", O.median(g.gradients))):
    got = bm.gars.__dict__[rule](dev, g.f).cpu().double()
    want = g.tensor(rule).double()
    keep = g.n - g.f
    win, amb, alts = _window_candidates(torch.nan_to_num(st, nan=math.inf), keep, centre)
    nan_centre = torch.isnan(centre)
    ok = ((got - want).abs() <= 2e-6 * scale) | (torch.isnan(got) & torch.isnan(want))
    # ambiguous columns: ours must be one of the legal windows
    legal = torch.zeros_like(ok)
    for alt in alts:
      legal |= (got - alt).abs() <= 2e-6 * scale
    ok |= amb & legal
    # NaN centre: documented deviation — we return NaN, the reference an arbitrary subset mean
    ok |= nan_centre & torch.isnan(got)
    assert bool(ok.all()), (rule, int((~ok).sum()))
    frac_amb = float(amb.float().mean())
    print(f"{name} {rule}: {int(amb.sum())} tie columns ({frac_amb:.2%}), {int(nan_centre.sum())} NaN-centre columns "
          f"of {st.shape[1]}")
    if not name.startswith("nan"):
      assert int(nan_centre.sum()) == 0


def test_closest_rules_full_size(bm):
  """phocas / meamed at n = 25, d = 11 173 962 against a float64 window formulation on the GPU."""
  n, f, d = 25, 5, D_RESNET18
  rows, h = gpu_stack("hetero", n, f, d, seed=11)
  keep = n - f
  st = torch.stack(rows)
  srt = st.sort(dim=0).values
  del st
  for rule in ("meamed", "phocas"):
    got = bm.gars.__dict__[rule](rows, f)
    centre = srt[(n - 1) // 2] if rule == "meamed" else srt[f:n - f].mean(dim=0)
    dev = (srt - centre).abs()
    # window start = number of leading values farther than their mirror (trmean.py:45-50 keeps the nearest)
    lo = torch.zeros(d, dtype=torch.long, device=DEV)
    hi = torch.full((d,), n - 1, dtype=torch.long, device=DEV)
    cols = torch.arange(d, device=DEV)
    for _ in range(n - keep):
      drop_lo = dev[lo, cols] > dev[hi, cols]
      lo = torch.where(drop_lo, lo + 1, lo)
      hi = torch.where(drop_lo, hi, hi - 1)
    total = torch.zeros(d, dtype=torch.float64, device=DEV)
    for k in range(keep):
      total += srt[lo + k, cols]
    want = total / keep
    # near ties at the window edge: our centre may differ from torch's in the last bit (phocas), which
    # legitimately flips the choice between two values that are equally far within rounding
    eps = 1e-6 * float(srt.abs().max())
    tie = ((dev[(lo - 1).clamp(min=0), cols] - dev[hi, cols]).abs() <= eps) & (lo > 0) | \
          ((dev[(hi + 1).clamp(max=n - 1), cols] - dev[lo, cols]).abs() <= eps) & (hi < n - 1)
    bad = ((got.double() - want).abs() > 2e-6 * float(srt.abs().max())) & ~tie
    assert int(bad.sum()) == 0, (rule, int(bad.sum()))
    assert int(tie.sum()) <= d // 1000, (rule, "tie columns", int(tie.sum()))


# ---------------------------------------------------------------------------- #
# The plugin objects, driven the way the reference's aggregators package drives them
# (aggregators/__init__.py:42-86, krum.py:82-96,159-166, median.py:41-49,80-87), on the GPU.

def _fake_aggregators_package():
  """A stand-in for the reference's `aggregators` package (absent on the GPU box): the registry
  contract only — `gars`, `register(name, unchecked, check, upper_bound=None, influence=None)` producing
  callables with the attributes check/checked/unchecked/upper_bound/influence — and the four
  native call sites with the reference's positional conventions."""
  pkg = types.ModuleType("aggregators")
  pkg.gars = {}

  def register(name, unchecked, check, upper_bound=None, influence=None):
    if name in pkg.gars:
      return
    def checked(**kwargs):
      message = check(**kwargs)
      if message is not None:
        raise RuntimeError(f"rule {name!r} rejected its parameters: {message}")
      return unchecked(**kwargs)
    for key, val in (("check", check), ("checked", checked), ("unchecked", unchecked), ("upper_bound", upper_bound),
                     ("influence", influence)):
      setattr(checked, key, val)
    pkg.gars[name] = checked
  pkg.register = register
  sys.modules["aggregators"] = pkg
  sys.modules.pop("native", None)
  import native  # registers native-trmean/-phocas/-meamed/-aksel/-average/-cge through pkg.register

  def accept(gradients, f=None, **kwargs):
    return None if isinstance(gradients, list) and len(gradients) >= 1 else "need a non-empty list"
  # the four call sites of the reference, positional arguments as written there
  register("native-krum", lambda gradients, f, m=None, **kw: native.krum.aggregate(
    gradients, f, len(gradients) - f - 2 if m is None else m), accept)
  register("native-bulyan", lambda gradients, f, m=None, **kw: native.bulyan.aggregate(
    gradients, f, len(gradients) - f - 2 if m is None else m), accept)
  register("native-median", lambda gradients, **kw: native.median.aggregate(gradients), accept)
  register("native-brute", lambda gradients, f, **kw: native.brute.aggregate(gradients, f), accept)
  return pkg, native


def test_plugin_objects_on_gpu(bm):
  saved = {k: sys.modules.get(k) for k in ("aggregators", "native")}
  try:
    pkg, native = _fake_aggregators_package()
    gars = pkg.gars
    for name in ("native-trmean", "native-phocas", "native-meamed", "native-aksel", "native-average", "native-cge",
                 "native-krum", "native-bulyan", "native-median", "native-brute"):
      assert name in gars, name
    g = Golden("hetero_n11_f2")
    dev = to_dev(g.gradients)
    model = object()  # rules must tolerate unknown keyword arguments (aggregators/__init__.py:15-21)
    assert same_bits(gars["native-median"](gradients=dev, f=g.f, model=model), g.tensor("median"))
    out = gars["native-trmean"](gradients=dev, f=g.f, model=model)
    assert float((out.cpu() - g.tensor("trmean")).abs().max()) <= 1e-6 * float(g.tensor("trmean").abs().max())
    assert all(out.data_ptr() != t.data_ptr() for t in dev)       # never an alias of an input
    assert same_bits(gars["native-krum"](gradients=dev, f=g.f, model=model), g.tensor("krum"))
    assert same_bits(gars["native-krum"](gradients=dev, f=g.f, m=1), g.tensor("krum_m1"))
    assert same_bits(native.krum.aggregate(dev, g.f, g.n - g.f - 2), g.tensor("krum"))  # positional, as krum.py:96
    assert same_bits(gars["native-brute"](gradients=dev, f=g.f), g.tensor("brute"))
    assert same_bits(gars["native-aksel"](gradients=dev, f=g.f), g.tensor("aksel_mid"))
    assert same_bits(gars["native-aksel"](gradients=dev, f=g.f, mode="n-f"), g.tensor("aksel_n-f"))
    assert same_bits(gars["native-average"](gradients=dev, f=g.f), g.tensor("average"))
    assert same_bits(gars["native-cge"](gradients=dev, f=g.f), g.tensor("cge"))
    g25 = Golden("hetero_n25_f5")
    dev25 = to_dev(g25.gradients)
    scale = float(torch.stack(g25.honests).abs().max())
    out = gars["native-bulyan"](gradients=dev25, f=g25.f, model=model)
    assert float((out.cpu() - g25.tensor("bulyan")).abs().max()) <= 2e-6 * scale
    # influence: late-bound on the reference-registered native-krum/-brute after their first call,
    # registered directly for the rules `native` registers itself
    order = g.array("krum_order").tolist()
    m = g.n - g.f - 2
    want = sum(1 for i in order[:m] if i >= g.h) / m
    assert gars["native-krum"].influence is not None
    assert gars["native-krum"].influence(dev[:g.h], dev[g.h:], f=g.f) == want
    sel = g.array("brute_selection").tolist()
    assert gars["native-brute"].influence(dev[:g.h], dev[g.h:], f=g.f) == sum(1 for i in sel if i >= g.h) / len(sel)
    ao = g.array("aksel_order").tolist()
    c = (g.n + 1) // 2
    assert gars["native-aksel"].influence(dev[:g.h], dev[g.h:], f=g.f) == sum(1 for i in ao[:c] if i >= g.h) / c
    assert gars["native-average"].influence(dev[:g.h], dev[g.h:], f=g.f) == (g.n - g.h) / g.n
    # checks are the callables handed to register(); a bad f must be refused by `checked`
    with pytest.raises(Exception):
      gars["native-trmean"].checked(gradients=dev, f=0)
    # errors: CPU tensors are refused loudly, there is no fallback
    with pytest.raises(bm.gars.GarInputError):
      gars["native-median"](gradients=[t.cpu() for t in dev], f=g.f)
  finally:
    for k, v in saved.items():
      if v is None:
        sys.modules.pop(k, None)
      else:
        sys.modules[k] = v


# ---------------------------------------------------------------------------- #
# The simulation step (attack.py:757-878 mirror): every momentum placement, clipping, both attacks,
# against the independent oracle loop of oracle/step_oracle.py

STEP_CONFIGS = [
  dict(gar="krum", momentum_at="worker", clip=None, attack="empire", factor=1.1),
  dict(gar="krum", momentum_at="worker", clip=100.0, attack="empire", factor=1.1),
  dict(gar="bulyan", momentum_at="server", clip=None, attack="little", factor=1.5),
  dict(gar="median", momentum_at="update", clip=105.0, attack="empire", factor=1.1),
  dict(gar="trmean", momentum_at="server", clip=95.0, attack="little", factor=-1.5),
  dict(gar="aksel", momentum_at="update", clip=None, attack="empire", factor=1.1),
  dict(gar="brute", momentum_at="worker", clip=None, attack="empire", factor=1.1),
  dict(gar="cge", momentum_at="worker", clip=None, attack="empire", factor=1.1),
]


@pytest.mark.parametrize("cfg", STEP_CONFIGS, ids=lambda c: f"{c['gar']}-{c['momentum_at']}-clip{c['clip']}-{c['attack']}")
def test_step_all_placements_against_reference_loop(bm, cfg):
  from byzantinemomentum_amd.step import AggregationStep
  from tests.step_reference import ReferenceLoop, assert_floats_close
  n, f, d = 11, 2, 30011
  h = n - f
  step = AggregationStep(n, f, f, gar=cfg["gar"], momentum=0.9, dampening=0.9, momentum_at=cfg["momentum_at"],
                         attack=cfg["attack"], attack_factor=cfg["factor"], nb_past=3, gradient_clip=cfg["clip"])
  ref = ReferenceLoop(n, f, f, cfg["gar"], cfg["momentum_at"], 0.9, 0.9, cfg["attack"], cfg["factor"], cfg["clip"], 3)
  gen = torch.Generator().manual_seed(123)
  origin = torch.randn(d, generator=gen)
  params = origin.clone()
  for it in range(4):
    base = 0.2 * torch.randn(d, generator=gen)
    sampled = [base + (0.5 + 0.1 * i) * torch.randn(d, generator=gen) for i in range(h + (1 if it == 2 else 0))]
    want_def, want_upd, want = ref.step(sampled, params, origin)
    got_def = step.run([g.to(DEV) for g in sampled], params.to(DEV), origin.to(DEV))
    scale = float(torch.stack(sampled).abs().max())
    assert float((got_def.cpu() - want_def).abs().max()) <= 4e-6 * scale, (cfg, it)
    assert float((step.update_gradient().cpu() - want_upd).abs().max()) <= 4e-6 * scale, (cfg, it)
    if it != 1:  # floats() skipped once: the past-gradient deque must advance regardless
      got = step.floats()
      assert step.floats() is got
      assert_floats_close(got, want, tag=(cfg["gar"], it), tol=1e-5)
    params = params - 0.05 * want_upd


# The factor search of the attacks (attacks/identical.py:67-77; the reference's default is factor=-16):
# scalar form (one distance pass over h+2 rows, then host only) and per-evaluation form
SEARCH_CONFIGS = [
  dict(gar="krum", momentum_at="worker", clip=None, attack="empire", evals=16),
  dict(gar="krum", momentum_at="worker", clip=None, attack="little", evals=16, line_search="generic"),
  dict(gar="krum", momentum_at="server", clip=100.0, attack="little", evals=9, negative=True),
  dict(gar="brute", momentum_at="update", clip=None, attack="little", evals=8),
  dict(gar="average", momentum_at="worker", clip=None, attack="empire", evals=5, negative=True),
  dict(gar="median", momentum_at="update", clip=None, attack="empire", evals=16),
  dict(gar="bulyan", momentum_at="server", clip=None, attack="little", evals=6),
  dict(gar="trmean", momentum_at="worker", clip=None, attack="little", evals=7),
]


@pytest.mark.parametrize("cfg", SEARCH_CONFIGS, ids=lambda c: f"{c['gar']}-{c['momentum_at']}-{c['attack']}-search{c['evals']}"
                                                              f"{'neg' if c.get('negative') else ''}-{c.get('line_search', 'auto')}")
def test_step_with_factor_search_against_reference_loop(bm, cfg):
  from byzantinemomentum_amd.step import AggregationStep
  from tests.step_reference import ReferenceLoop, assert_floats_close
  n, f, d = 11, 2, 30011
  h = n - f
  step = AggregationStep(n, f, f, gar=cfg["gar"], momentum=0.9, dampening=0.9, momentum_at=cfg["momentum_at"],
                         attack=cfg["attack"], nb_past=3, gradient_clip=cfg["clip"], attack_evals=cfg["evals"],
                         attack_negative=cfg.get("negative", False), line_search=cfg.get("line_search", "auto"))
  assert not step.single_call
  ref = ReferenceLoop(n, f, f, cfg["gar"], cfg["momentum_at"], 0.9, 0.9, cfg["attack"], 1.1, cfg["clip"], 3,
                      evals=cfg["evals"], negative=cfg.get("negative", False))
  gen = torch.Generator().manual_seed(321)
  origin = torch.randn(d, generator=gen)
  params = origin.clone()
  for it in range(3):
    base = 0.2 * torch.randn(d, generator=gen)
    sampled = [base + (0.5 + 0.1 * i) * torch.randn(d, generator=gen) for i in range(h)]
    want_def, want_upd, want = ref.step(sampled, params, origin)
    got_def = step.run([g.to(DEV) for g in sampled], params.to(DEV), origin.to(DEV))
    got_search, want_search = step.last_search, ref.last_search
    assert len(got_search) == len(want_search) == cfg["evals"]
    # an objective that is zero up to rounding (the rule returned the honest mean): the scalar form gets it from a
    # cancellation among h^2 inner products, so the absolute floor is relative to the spread of the honest rows
    floor = 1e-8 * sum(v * v for v in [want["honest_norm_dev"]]) * h
    # the same candidates in the same order, the same objective at each: the same decisions, the same factor
    for (x, y), (xo, yo) in zip(got_search, want_search):
      assert x == xo and abs(y - yo) <= 2e-5 * abs(yo) + floor, (cfg, it, x, y, yo)
    assert step.last_factor == ref.last_factor, (cfg, it)
    scale = float(torch.stack(sampled).abs().max()) * max(1.0, abs(ref.last_factor))
    assert float((got_def.cpu() - want_def).abs().max()) <= 4e-6 * scale, (cfg, it)
    assert_floats_close(step.floats(), want, tag=(cfg["gar"], it), tol=1e-5)
    params = params - 0.05 * want_upd


@pytest.mark.parametrize("kind,attack", [("hetero", "empire"), ("tight", "little")])
def test_factor_search_full_size_c3_both_forms_and_fp64(bm, kind, attack):
  """The search against Multi-Krum at C3 (n=51, f=12, d=11 173 962): the scalar form (one distance pass) and the
  per-evaluation form (the HIP rule on the vectors, sixteen times) visit the same candidates with the same
  objective and settle on the same factor; the objective of the final factor is checked against a float64
  evaluation on the GPU of what identical.py:72-76 computes."""
  from byzantinemomentum_amd.step import AggregationStep
  n, f, d = 51, 12, D_RESNET18
  rows, h = gpu_stack(kind, n, f, d, seed=31)
  honests = rows[:h]
  avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack=attack, direction=True)
  found = {}
  for mode in ("auto", "generic"):
    runner = AggregationStep(n, f, f, gar="krum", attack=attack, attack_evals=16, line_search=mode, nb_past=0)
    runner.last_factor = runner._search_factor(honests, avg, direction)  # ("auto": the device search's tensor)
    found[mode] = (runner.last_factor, runner.last_search)
  (fa, ta), (fg, tg) = found["auto"], found["generic"]
  top = max(y for _, y in tg)
  for (x, y), (xo, yo) in zip(ta, tg):
    assert x == xo and abs(y - yo) <= 1e-5 * abs(yo) + 1e-9 * top, (x, y, yo)
  assert fa == fg
  # float64 on the GPU: candidate vector as the reference rounds it, Multi-Krum of the 51 rows, distance to avg
  cand = avg + fa * direction
  grads = honests + [cand] * f
  sel = bm.gars.krum_selection(grads, f)
  acc = torch.zeros(d, dtype=torch.float64, device=DEV)
  for i in sel:
    acc += grads[i].double()
  want = float((acc / len(sel) - avg.double()).pow(2).sum())
  got = dict(ta)[fa]
  assert abs(got - want) <= 1e-5 * want, (got, want)


def test_selected_mean_burst_form_at_short_lengths():
  """bm_selected_mean has a burst form (one workgroup per CU, results staged in LDS) used from 8 iterations per CU
  on; BM_MEAN_BURST=1 sends gradients of one to two million coordinates with a ragged tail through it (a full and
  a partial iteration, d % 4 != 0), where scripts/selected_mean_probe.py compares the WHOLE result with torch's own
  sequential adds, bit for bit.  The knob is read once per process: subprocess."""
  env = dict(os.environ, BM_MEAN_BURST="1", PYTHONPATH=ROOT, BM_PROBE_CASES="16:13:1310723,64:37:1100003,25:18:2621443")
  out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "selected_mean_probe.py")], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=600)
  assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
  lines = [ln for ln in out.stdout.splitlines() if ln.startswith("BM_MEAN_BURST=1")]
  assert len(lines) == 3 and all("bit-exact True" in ln for ln in lines), out.stdout[-2000:]


def test_attack_direction_output(bm):
  """BM_ATTACK_DIRECTION: the attack direction alone (grad_att of identical.py:65) from bm_stack_stats and both
  forms of bm_momentum_stats, bit-identical to byz - avg being rebuilt the reference's way."""
  gen = torch.Generator().manual_seed(17)
  for h, d in ((7, 4099), (20, 30011), (39, 5003)):
    rows = [torch.randn(d, generator=gen) for _ in range(h)]
    for attack in ("empire", "little"):
      dev = [g.to(DEV) for g in rows]
      avg, _, direction = bm.stats.stack_stats_async(dev, scale=1.0, attack=attack, direction=True)
      _, _, byz = bm.stats.stack_stats_async(dev, scale=1.0, attack=attack)
      assert same_bits(byz, avg.cpu() + direction.cpu())
      stck = torch.stack(rows)
      want = stck.mean(dim=0).neg() if attack == "empire" else stck.var(dim=0).sqrt_()
      assert float((direction.cpu() - want).abs().max()) <= 4e-6 * float(want.abs().max())
      bufs = [torch.zeros(d, device=DEV) for _ in range(h)]
      _, h_avg, direction2, _ = bm.stats.momentum_stats(dev, bufs, 0.0, 1.0, None, 1.0, attack, direction=True)
      assert float((direction2.cpu() - want).abs().max()) <= 4e-6 * float(want.abs().max())
      bufs = [torch.zeros(d, device=DEV) for _ in range(h)]
      _, h_avg2, byz2, _ = bm.stats.momentum_stats(dev, bufs, 0.0, 1.0, None, 1.0, attack)
      assert same_bits(byz2, h_avg2.cpu() + direction2.cpu())


def test_momentum_stats_kernel_tiers(bm):
  """bm_momentum_stats for row counts in every register tier (<= 8, 12, 20, 40, 64), odd lengths,
  ks > h, clipping factors, both attacks; bit-exact momentum, averages and Byzantine vector."""
  gen = torch.Generator().manual_seed(9)
  for ks, h, d in ((3, 3, 1001), (8, 7, 4099), (12, 12, 2050), (20, 20, 30011), (25, 20, 10007), (39, 39, 5003),
                   (64, 50, 2049)):
    sampled = [torch.randn(d, generator=gen) for _ in range(ks)]
    bufs = [torch.randn(d, generator=gen) for _ in range(h)]
    factors = torch.ones(64)
    factors[1] = 0.5
    factors[ks - 1] = 0.25
    for attack, scale in (("empire", 1.1), ("little", -1.5)):
      dbufs = [b.to(DEV) for b in bufs]
      s_avg, h_avg, byz, out6 = bm.stats.momentum_stats([g.to(DEV) for g in sampled], dbufs, 0.9, 0.1, factors.to(DEV),
                                                        scale, attack)
      clipped = [g * factors[i] for i, g in enumerate(sampled)]
      want_bufs = [b.clone().mul_(0.9).add_(g, alpha=0.1) for b, g in zip(bufs, clipped)]
      for a, b in zip(dbufs, want_bufs):
        assert float((a.cpu() - b).abs().max()) <= 1e-6 * float(b.abs().max())
      ws_avg, ws_norm, ws_dev, ws_max = O.compute_avg_dev_max(clipped, "f64")
      wh_avg, wh_norm, wh_dev, wh_max = O.compute_avg_dev_max(want_bufs, "f64")
      assert same_bits(s_avg, O.compute_avg_dev_max(clipped)[0])
      assert float((h_avg.cpu() - O.compute_avg_dev_max(want_bufs)[0]).abs().max()) <= 2e-7 * wh_max
      o = out6.tolist()
      for got, want in ((math.sqrt(o[0]), ws_norm), (math.sqrt(o[1] / (ks - 1)), ws_dev), (o[2], ws_max),
                        (math.sqrt(o[3]), wh_norm), (math.sqrt(o[4] / (h - 1)), wh_dev), (o[5], wh_max)):
        assert abs(got - want) <= 1e-5 * want, (ks, h, d, attack, got, want)
      stck = torch.stack(want_bufs)
      avg = stck.mean(dim=0)
      att = avg.neg() if attack == "empire" else stck.var(dim=0).sqrt_()
      want_byz = avg + scale * att
      assert float((byz.cpu() - want_byz).abs().max()) <= 4e-6 * float(want_byz.abs().max()), (ks, h, attack)


def test_momentum_stats_other_form():
  """bm_momentum_stats has two forms (register-resident two-pass up to 20 rows, streaming with a pivot
  above; BM_STEP_STREAM=1 forces the streaming form at every size); the library reads the knob once per
  process, so the forced form runs in a subprocess."""
  other = "1" if os.environ.get("BM_STEP_STREAM", "0") != "1" else "0"
  env = dict(os.environ, BM_STEP_STREAM=other, PYTHONPATH=ROOT)
  out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity_r2.py"), "-q", "-x",
                        "-m", "gpu", "-k", "test_momentum_stats_kernel_tiers or test_step_all_placements or test_attack_direction_output"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
  assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])


def test_full_size_c5_step_against_fp64(bm):
  """BASELINE config 5 on one GPU: one step at d = 36 546 980 (WRN-28-10 / CIFAR-100), n = 25, f = 5,
  worker momentum 0.99, empire 1.1, Krum, against float64 reductions computed with torch on the same GPU."""
  from byzantinemomentum_amd.step import AggregationStep
  n, f, d, mu, damp = 25, 5, D_WRN, 0.99, 0.99
  h = n - f
  gen = torch.Generator(device=DEV).manual_seed(77)
  mu_vec = 0.1 * torch.randn(d, device=DEV, generator=gen)
  step = AggregationStep(n, f, f, gar="krum", momentum=mu, dampening=damp, attack_factor=1.1, nb_past=25)
  ref_bufs = [torch.zeros(d, device=DEV) for _ in range(h)]
  past = None
  for it in range(2):
    sampled = [mu_vec + s * torch.randn(d, device=DEV, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()]
    defense = step.run(sampled)
    got = step.floats()
    # reference on the same GPU: torch's own fp32 momentum, float64 for everything that is reduced
    for b, g in zip(ref_bufs, sampled):
      b.mul_(mu).add_(g, alpha=1.0 - damp)
    for a, b in zip(step.buffers, ref_bufs):
      assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())

    def stats64(rows):
      avg = rows[0].clone()
      for r in rows[1:]:
        avg.add_(r)
      avg.div_(len(rows))
      a64 = avg.double()
      dev2 = sum(float((r.double() - a64).pow(2).sum()) for r in rows)
      return avg, math.sqrt(float(a64.pow(2).sum())), math.sqrt(dev2 / (len(rows) - 1)) if len(rows) > 1 else math.nan, \
          float(avg.abs().max())
    s_avg, s_norm, s_dev, s_max = stats64(sampled)
    h_avg, h_norm, h_dev, h_max = stats64(ref_bufs)
    byz = h_avg + 1.1 * (-h_avg)
    d_norm = math.sqrt(float(defense.double().pow(2).sum()))
    for key, want in (("sampled_norm_avg", s_norm), ("sampled_norm_dev", s_dev), ("sampled_norm_max", s_max),
                      ("honest_norm_avg", h_norm), ("honest_norm_dev", h_dev), ("honest_norm_max", h_max),
                      ("defense_norm_avg", d_norm), ("defense_norm_max", float(defense.abs().max())),
                      ("attack_norm_avg", math.sqrt(float(byz.double().pow(2).sum())))):
      assert abs(got[key] - want) <= 1e-5 * want, (it, key, got[key], want)

    def cos64(a, b):
      return float(torch.dot(a.double(), b.double())) / math.sqrt(float(a.double().pow(2).sum())) / \
          math.sqrt(float(b.double().pow(2).sum()))
    for key, want in (("cosin_splhon", cos64(s_avg, h_avg)), ("cosin_spldef", cos64(s_avg, defense)),
                      ("cosin_hondef", cos64(h_avg, defense)), ("cosin_splatt", cos64(s_avg, byz)),
                      ("cosin_attdef", cos64(byz, defense))):
      assert abs(got[key] - want) <= 1e-5, (it, key, got[key], want)
    if past is not None:
      assert abs(got["cosin_sampled"] - cos64(s_avg, past)) <= 1e-5
      want_curv = mu * float(torch.dot(s_avg.double(), past.double()))
      assert abs(got["curv_sampled"] - want_curv) <= 1e-5 * max(abs(want_curv), 1.0)
    # the rule itself: selection from float64 distances on the same GPU
    grads = list(ref_bufs) + [byz] * f
    scores = O.krum_scores(np.sqrt(sqdist_f64_on_gpu(grads)), f)
    order = O._stable_order(scores)
    m = n - f - 2
    acc = torch.zeros(d, dtype=torch.float32, device=DEV)
    for i in order[:m]:
      acc = acc + grads[i]
    want_def = acc / m
    assert float((defense - want_def).abs().max()) <= 4e-6 * float(want_def.abs().max())
    past = s_avg


# ---------------------------------------------------------------------------- #
# The NaN attack (attacks/nan.py: f all-NaN gradients) through every rule, against the real reference's outputs

def test_nan_attack_through_every_rule(bm):
  g = Golden("nan_n11_f2")
  dev = to_dev(g.gradients)
  scale = float(torch.stack(g.honests).abs().max())
  assert same_bits(bm.median(dev), g.tensor("median"))                 # NaN everywhere: torch's median propagates
  assert bool(torch.isnan(g.tensor("median")).all())
  assert same_bits(bm.meamed(dev, g.f), g.tensor("meamed"))             # NaN centre -> NaN, as the reference returns here
  assert float((bm.trmean(dev, g.f).cpu() - g.tensor("trmean")).abs().max()) <= 1e-6 * scale
  assert float((bm.phocas(dev, g.f).cpu() - g.tensor("phocas")).abs().max()) <= 2e-6 * scale   # no column exempt
  assert same_bits(bm.krum(dev, g.f), g.tensor("krum"))
  assert float((bm.bulyan(dev, g.f).cpu() - g.tensor("bulyan")).abs().max()) <= 2e-6 * scale
  assert same_bits(bm.average(dev), g.tensor("average"))
  avg, norm, devi, mx = bm.compute_avg_dev_max(dev[g.h:])                # statistics of the all-NaN attack stack
  assert math.isnan(norm) and math.isnan(mx)


def test_zero_length_inputs_through_every_entry_point(bm):
  """d = 0 (an empty trailing shard of a sharded job): every entry point must succeed and write
  neutral values, so that every rank reaches its collectives."""
  from byzantinemomentum_amd.sharded import ShardedAggregator
  n, f = 7, 1
  rows = [torch.zeros(0, device=DEV) for _ in range(n)]
  assert bool((bm.gars.pairwise_sqdist(rows) == 0).all())
  avg, out3 = bm.stats.stack_stats_async(rows)
  assert avg.shape == (0,) and out3.tolist() == [0.0, 0.0, 0.0]
  gram, ex = bm.stats.study_dots(rows[:3], rows[3:5])
  assert bool((gram == 0).all()) and bool((ex == 0).all())
  s_avg, h_avg, byz, out6 = bm.stats.momentum_stats(rows, [torch.zeros(0, device=DEV) for _ in range(n)], 0.9, 0.1,
                                                    None, 1.1, "empire")
  assert out6.tolist() == [0.0] * 6 and byz.shape == (0,)
  for rule in (bm.krum, bm.bulyan, bm.trmean, bm.aksel, bm.cge):
    assert rule(rows, f).shape == (0,)
  agg = ShardedAggregator()
  assert agg.krum(rows, f).shape == (0,) and agg.bulyan(rows, f).shape == (0,)
  assert bm.gars.krum_selection(rows, f) == list(range(n - f - 2))      # all-zero distances: ties by index


# ---------------------------------------------------------------------------- #
# The single-call step (bm_step_worker) issues the same kernels as the Python sequence: identical bits

@pytest.mark.parametrize("gar,clip,attack,factor,f_real", [("krum", None, "empire", 1.1, 5), ("bulyan", 150.0, "little", 1.5, 5),
                                                         ("median", None, "empire", 1.1, 5), ("trmean", 140.0, "little", -1.5, 5),
                                                         ("meamed", None, "empire", 1.1, 0), ("phocas", None, "empire", 1.1, 3)])
def test_single_call_step_equals_python_sequence(bm, gar, clip, attack, factor, f_real):
  from byzantinemomentum_amd.step import AggregationStep
  n, f, d = 25, 5, 40013
  h = n - f_real
  kw = dict(gar=gar, momentum=0.9, dampening=0.9, attack=attack, attack_factor=factor, nb_past=3, gradient_clip=clip)
  one = AggregationStep(n, f, f_real, single_call=True, **kw)
  seq = AggregationStep(n, f, f_real, single_call=False, **kw)
  assert one.single_call and not seq.single_call
  gen = torch.Generator(device=DEV).manual_seed(31)
  origin = torch.randn(d, device=DEV, generator=gen)
  params = origin + 0.01
  for it in range(5):
    base = 0.2 * torch.randn(d, device=DEV, generator=gen)
    sampled = [base + (0.5 + 0.05 * i) * torch.randn(d, device=DEV, generator=gen) for i in range(h + (1 if it == 3 else 0))]
    a = one.run([g.clone() for g in sampled], params, origin)
    b = seq.run([g.clone() for g in sampled], params, origin)
    assert torch.equal(a, b), (gar, it)
    for x, y in zip(one.buffers, seq.buffers):
      assert torch.equal(x, y)
    if it != 1:
      fa, fb = one.floats(), seq.floats()
      for key in fb:
        assert fa[key] == fb[key] or (math.isnan(fa[key]) and math.isnan(fb[key])), (gar, it, key, fa[key], fb[key])
    params = params - 0.05 * a
