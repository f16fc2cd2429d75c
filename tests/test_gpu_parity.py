"""Parity of the HIP path (through the C ABI and the host mirror) with the oracle / the golden
fixtures of the real reference.  Needs an MI355X: `pytest -m gpu`.

Bars (BASELINE.json north_star): selected index sets bit-identical; median bit-exact; means and
statistics within 1e-5 — tightened here to what is achievable: sequential means are bit-exact,
sorted-order means within 1e-6 (summation order only).
"""

import math
import os
import subprocess
import sys

import pytest
import torch

from oracle import gar_oracle as O
from tests.golden_io import CASES, HAND_CASES, Golden, same_bits

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bm():
  import byzantinemomentum_amd
  byzantinemomentum_amd._lib.load()  # must be the in-tree HIP library, loudly
  return byzantinemomentum_amd


def close(x, ref, tol, scale=None):
  """|x - ref| <= tol * max(|ref|, scale) elementwise, NaN matching NaN."""
  x = torch.as_tensor(x).detach().cpu().to(torch.float64)
  ref = torch.as_tensor(ref).detach().cpu().to(torch.float64)
  nx, nr = torch.isnan(x), torch.isnan(ref)
  if not bool((nx == nr).all()):
    return False
  if scale is None:
    scale = float(ref[~nr].abs().max()) if (~nr).any() else 1.0
  bound = tol * torch.maximum(ref.abs(), torch.tensor(scale, dtype=torch.float64))
  return bool(((x - ref).abs()[~nr] <= bound[~nr]).all())


def to_dev(gradients):
  """Move a list to the GPU keeping the aliasing structure (same object -> same object)."""
  seen = {}
  out = []
  for g in gradients:
    if id(g) not in seen:
      seen[id(g)] = g.to(DEV)
    out.append(seen[id(g)])
  return out


# ---------------------------------------------------------------------------- #
# Golden fixtures of the real reference

@pytest.mark.parametrize("name", CASES + HAND_CASES)
def test_golden_colwise(bm, name):
  g = Golden(name)
  dev = to_dev(g.gradients)
  assert same_bits(bm.median(dev), g.tensor("median"))  # bit-exact, NaN columns included
  if g.has("trmean"):
    assert close(bm.trmean(dev, g.f), g.tensor("trmean"), 1e-6)
  # phocas / meamed on the same fixtures: tests/test_gpu_parity_r2.py::test_closest_rules_without_exemptions
  # (no column is exempt: an exact deviation tie must give a legal window)


@pytest.mark.parametrize("name", CASES)
def test_golden_krum(bm, name):
  g = Golden(name)
  dev = to_dev(g.gradients)
  m = g.n - g.f - 2
  want_order = g.array("krum_order").tolist()
  got_sel = bm.gars.krum_selection(dev, g.f)
  if name.startswith("iid"):
    # iid rows: scores within rounding of each other, the order is ill-conditioned;
    # require agreement with the float64 oracle instead, gated by the decisive gap
    order64, scores64 = O.krum_order(g.gradients, g.f, "f64")
    srt = sorted(scores64)
    if srt[m] - srt[m - 1] > 1e-6 * srt[m]:
      assert sorted(got_sel) == sorted(order64[:m])
  else:
    assert got_sel == want_order[:m]  # bit-identical selection, in score order
    assert same_bits(bm.krum(dev, g.f), g.tensor("krum"))        # hence a bit-identical average
    assert same_bits(bm.krum(dev, g.f, 1), g.tensor("krum_m1"))
    # scores: ours use exact distances, the reference's fp32 norm is ~1e-6 off at this d
    _, scores = bm.gars._rank(dev, g.f, m, bm._lib.RANK_KRUM)
    got_scores = [scores[i].item() for i in want_order]
    for a, b in zip(got_scores, g.array("krum_scores").tolist()):
      assert (math.isinf(a) and math.isinf(b)) or abs(a - b) <= 1e-5 * abs(b)


@pytest.mark.parametrize("name", [c for c in CASES if Golden(c).has("bulyan") and not c.startswith("iid")])
def test_golden_bulyan(bm, name):
  g = Golden(name)
  dev = to_dev(g.gradients)
  ref_order, _ = O.bulyan_order(g.gradients, g.f)
  assert bm.gars.bulyan_ranking(dev, g.f) == ref_order
  scale = float(torch.stack(g.honests).abs().max())
  assert close(bm.bulyan(dev, g.f), g.tensor("bulyan"), 2e-6, scale)
  assert close(bm.bulyan(dev, g.f, 3), g.tensor("bulyan_m3"), 2e-6, scale)


@pytest.mark.parametrize("name", [c for c in CASES if not c.startswith("iid")])
def test_golden_brute_aksel_average_cge(bm, name):
  g = Golden(name)
  dev = to_dev(g.gradients)
  if g.has("brute"):
    assert bm.gars.brute_selection(dev, g.f) == g.array("brute_selection").tolist()
    assert same_bits(bm.brute(dev, g.f), g.tensor("brute"))
  if not name.startswith("nan"):  # NaN distances: the reference's own order is unspecified (Python sort of NaN)
    assert bm.gars.aksel_selection(dev, g.f, "n-f") == g.array("aksel_order").tolist()[:g.n - g.f]
    assert same_bits(bm.aksel(dev, g.f, "mid"), g.tensor("aksel_mid"))
    assert same_bits(bm.aksel(dev, g.f, "n-f"), g.tensor("aksel_n-f"))
    assert same_bits(bm.cge(dev, g.f), g.tensor("cge"))
  assert same_bits(bm.average(dev), g.tensor("average"))


@pytest.mark.parametrize("name", CASES)
def test_golden_stats(bm, name):
  g = Golden(name)
  for prefix, samples in (("honest", g.honests), ("attack", g.attacks)):
    if not g.has(prefix + "_stats"):
      continue
    avg, norm, dev, mx = bm.compute_avg_dev_max(to_dev(samples))
    assert same_bits(avg, g.tensor(prefix + "_avg"))  # sequential mean: bit-exact
    want = g.array(prefix + "_stats").tolist()
    scale = want[0] if math.isfinite(want[0]) else 1.0
    for a, b in zip((norm, dev, mx), want):
      assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= 1e-5 * max(abs(b), scale), (prefix, a, b)


# ---------------------------------------------------------------------------- #
# Every row count the kernels are instantiated for, odd lengths, unaligned views

@pytest.mark.parametrize("n", list(range(1, 65)))
def test_every_n_median_trmean(bm, n):
  gen = torch.Generator().manual_seed(1000 + n)
  d = 2051  # not a multiple of 4: vector body + scalar tail
  rows = [torch.randn(d, generator=gen) for _ in range(n)]
  dev = [r.to(DEV) for r in rows]
  assert torch.equal(bm.median(dev).cpu(), torch.stack(rows).median(dim=0).values)
  f = (n - 1) // 2
  if f >= 1:
    for ff in sorted({1, f}):
      assert close(bm.trmean(dev, ff), O.trmean(rows, ff), 1e-6)
      st = torch.stack(rows)
      got = bm.meamed(dev, ff).cpu().double()
      win, amb = O.closest_window(st, n - ff, st.median(dim=0).values)
      assert bool((((got - win).abs() <= 2e-6 * float(st.abs().max())) | amb).all())


@pytest.mark.parametrize("offset", [0, 1, 2, 3])
def test_unaligned_rows_and_tails(bm, offset):
  """Rows that are views at 4/8/12-byte offsets (e.g. slices of one flat buffer) take the narrower
  vector paths; results must not change."""
  n, f, d = 13, 3, 4099
  gen = torch.Generator().manual_seed(77)
  flat = torch.randn(n * (d + 8), generator=gen).to(DEV)
  dev = [flat[i * (d + 8) + offset: i * (d + 8) + offset + d] for i in range(n)]
  rows = [t.cpu() for t in dev]
  assert torch.equal(bm.median(dev).cpu(), O.median(rows))
  assert close(bm.trmean(dev, f), O.trmean(rows, f), 1e-6)
  assert bm.gars.krum_selection(dev, f) == O.krum_order(rows, f, "f64")[0][:n - f - 2]
  assert torch.equal(bm.krum(dev, f).cpu(), O.krum(rows, f))
  assert torch.equal(bm.average(dev).cpu(), O.average(rows))
  avg, norm, devi, mx = bm.compute_avg_dev_max(dev)
  wavg, wnorm, wdev, wmx = O.compute_avg_dev_max(rows, "f64")
  assert torch.equal(avg.cpu(), O.compute_avg_dev_max(rows)[0])
  assert abs(norm - wnorm) <= 1e-6 * wnorm and abs(devi - wdev) <= 1e-6 * wdev and abs(mx - wmx) <= 1e-6 * wmx


def test_empty_and_tiny_lengths(bm):
  rows = [torch.zeros(0, device=DEV) for _ in range(5)]
  assert bm.median(rows).shape == (0,)
  assert bm.trmean(rows, 1).shape == (0,)
  one = [torch.tensor([float(i)], device=DEV) for i in (3, 1, 2, 5, 4)]
  assert bm.median(one).item() == 3.0 and bm.trmean(one, 1).item() == 3.0
  assert math.isfinite(bm.krum(one, 1).item())


def test_nan_semantics(bm):
  """torch 2.10: median propagates NaN; sort puts NaN last so trmean is NaN iff > f NaNs."""
  n, f, d = 9, 2, 1030
  gen = torch.Generator().manual_seed(5)
  rows = [torch.randn(d, generator=gen) for _ in range(n)]
  rows[1][::7] = math.nan
  rows[4][::7] = math.nan
  rows[6][::21] = math.nan        # 3 NaNs in every 21st column (> f)
  rows[2][5] = math.inf
  rows[3][5] = -math.inf
  dev = [r.to(DEV) for r in rows]
  assert same_bits(bm.median(dev), O.median(rows))
  got, want = bm.trmean(dev, f).cpu(), O.trmean(rows, f)
  assert bool((torch.isnan(got) == torch.isnan(want)).all())
  assert close(got, want, 1e-6)
  got, want = bm.phocas(dev, f).cpu(), O.phocas(rows, f)
  ok = torch.isnan(want) | ((got - want).abs() <= 1e-5)   # where the reference is finite we agree
  assert bool(ok.all())


# ---------------------------------------------------------------------------- #
# Seeded stacks at a moderate size against both oracle modes

@pytest.mark.parametrize("kind,n,f", [("hetero", 25, 5), ("little", 25, 5), ("hetero", 51, 12), ("hetero", 11, 2)])
def test_seeded_stack_100k(bm, kind, n, f):
  d = 100003 if n <= 25 else 30011  # the CPU oracle's pair loop is quadratic in n (full size: test_gpu_parity_r2.py)
  rows, h = O.make_stack(kind, n, f, d, seed=99)
  dev = to_dev(rows)
  m = n - f - 2
  # selections: identical to the reference-faithful f32 oracle AND the float64 truth
  o32, _ = O.krum_order(rows, f, "f32")
  o64, _ = O.krum_order(rows, f, "f64")
  sel = bm.gars.krum_selection(dev, f)
  assert sel == o32[:m] == o64[:m]
  assert torch.equal(bm.krum(dev, f).cpu(), O.krum(rows, f))
  if n >= 4 * f + 3:
    assert bm.gars.bulyan_ranking(dev, f) == O.bulyan_order(rows, f, None, "f64")[0]
    scale = float(torch.stack(rows[:h]).abs().max())
    assert close(bm.bulyan(dev, f), O.bulyan(rows, f), 2e-6, scale)
  assert bm.gars.aksel_selection(dev, f) == O.aksel_order(rows, "f64")[0][:(n + 1) // 2]
  assert torch.equal(bm.aksel(dev, f).cpu(), O.aksel(rows, f))
  assert torch.equal(bm.median(dev).cpu(), O.median(rows))
  assert close(bm.trmean(dev, f), O.trmean(rows, f), 1e-6)
  # squared distances against float64
  sq = bm.gars.pairwise_sqdist(dev).cpu()
  d64 = torch.from_numpy(O.pairwise_distances(rows, "f64")) ** 2
  assert close(sq, d64, 1e-6, float(d64.max()) * 1e-3)
  assert bool((sq.diagonal() == 0).all()) and torch.equal(sq, sq.T)
  # aliased Byzantine rows: exact zeros between them, bitwise-equal distances to everyone else
  for a in range(h + 1, n):
    assert sq[h, a].item() == 0.0
    assert torch.equal(sq[h, :h], sq[a, :h])


def test_study_block_and_momentum(bm):
  n, f, d = 25, 5, 50021
  rows, h = O.make_stack("hetero", n, f, d, seed=3)
  dev = to_dev(rows)
  gen = torch.Generator().manual_seed(8)
  pasts = [torch.randn(d, generator=gen) for _ in range(5)]
  defense = O.krum(rows, f)
  want = O.study_block(rows[:h], rows[:h], rows[h:], defense, [(p, p.norm().item()) for p in pasts], 0.9, "f64")
  s_avg, s_norm, s_dev, s_max = bm.compute_avg_dev_max(dev[:h])
  a_avg, a_norm, a_dev, a_max = bm.compute_avg_dev_max(dev[h:])
  gram, extra = bm.stats.study_dots([s_avg, a_avg, defense.to(DEV)], [p.to(DEV) for p in pasts])
  gram, extra = gram.cpu(), extra.cpu()
  assert abs(math.sqrt(gram[0, 0]) - want["sampled_norm_avg"]) <= 1e-6 * want["sampled_norm_avg"]
  assert abs(s_dev - want["sampled_norm_dev"]) <= 1e-6 * want["sampled_norm_dev"]
  assert abs(a_dev - want["attack_norm_dev"]) <= 1e-5 * max(want["attack_norm_dev"], a_norm)
  cos_sa = gram[0, 1].item() / math.sqrt(gram[0, 0].item()) / math.sqrt(gram[1, 1].item())
  assert abs(cos_sa - want["cosin_splatt"]) <= 1e-5
  cos_sd = gram[0, 2].item() / math.sqrt(gram[0, 0].item()) / math.sqrt(gram[2, 2].item())
  assert abs(cos_sd - want["cosin_spldef"]) <= 1e-5
  curv = 0.9 * sum(0.9 ** i * extra[i].item() for i in range(len(pasts)))
  assert abs(curv - want["curv_sampled"]) <= 1e-5 * max(abs(want["curv_sampled"]), 1.0)
  # worker momentum, in place, bit-exact against torch's fused mul_/add_
  bufs = [torch.randn(d, generator=gen) for _ in range(h)]
  dbufs = [b.to(DEV) for b in bufs]
  bm.stats.multi_axpby(dbufs, dev[:h], 0.99, 1.0 - 0.1)
  O.worker_momentum(bufs, rows[:h], 0.99, 0.1)
  for a, b in zip(dbufs, bufs):
    assert close(a, b, 1e-6)


def test_generic_bulyan_kernel_equals_specialised(bm, monkeypatch):
  """(n, f) outside the register-resident table and m != m_max go through the LDS kernel."""
  for n, f, m in ((13, 2, None), (25, 5, 7), (29, 6, None)):
    rows, h = O.make_stack("hetero", n, f, 3001, seed=21)
    dev = to_dev(rows)
    scale = float(torch.stack(rows[:h]).abs().max())
    assert close(bm.bulyan(dev, f, m), O.bulyan(rows, f, m), 2e-6, scale)


# ---------------------------------------------------------------------------- #
# Full size (BASELINE.json configs 2-4): size-independent properties + torch on the same GPU

@pytest.fixture(scope="module")
def full_stack():
  n, f, d = 25, 5, 11173962
  gen = torch.Generator(device=DEV).manual_seed(99)
  mu = 0.1 * torch.randn(d, device=DEV, generator=gen)
  h = n - f
  sig = torch.linspace(0.5, 1.5, h)
  honest = [mu + sig[i].item() * torch.randn(d, device=DEV, generator=gen) for i in range(h)]
  byz = torch.stack(honest).mean(dim=0).mul_(-0.1)
  return honest + [byz] * f, h, f


def test_full_size_colwise_properties(bm, full_stack):
  rows, h, f = full_stack
  n = len(rows)
  med = bm.median(rows)
  st = torch.stack(rows)
  assert torch.equal(med, st.median(dim=0).values)                 # same GPU, torch's own kernel
  tm = bm.trmean(rows, f)
  ref_tm = st.sort(dim=0).values[f:n - f].mean(dim=0)
  assert close(tm, ref_tm, 1e-6)
  del st, ref_tm
  perm = [rows[i] for i in torch.randperm(n).tolist()]
  assert torch.equal(bm.median(perm), med)                          # permutation invariance, bitwise
  assert torch.equal(bm.trmean(perm, f), tm)                        # sorted-order sum: also bitwise
  doubled = [r * 2 for r in rows[:h]] + [rows[h] * 2] * f
  assert torch.equal(bm.median(doubled), med * 2)                   # exact scaling by a power of two
  assert torch.equal(bm.trmean(doubled, f), tm * 2)
  assert torch.equal(bm.median([rows[3]] * n), rows[3])             # idempotence
  assert close(bm.trmean([rows[3]] * n, f), rows[3], 1e-6)          # (15 r)/15 rounds in fp32
  lo = torch.stack(rows).min(dim=0).values
  assert bool((med >= lo).all())


def test_full_size_krum_bulyan_properties(bm, full_stack):
  rows, h, f = full_stack
  n = len(rows)
  sq = bm.gars.pairwise_sqdist(rows).cpu()
  assert torch.equal(sq, sq.T) and bool((sq.diagonal() == 0).all())
  for a in range(h + 1, n):
    assert sq[h, a].item() == 0.0 and torch.equal(sq[h, :h], sq[a, :h])
  # a few entries against float64 on the GPU
  for (i, j) in ((0, 1), (3, 17), (7, 22), (19, 24)):
    want = (rows[i].double() - rows[j].double()).pow(2).sum().item()
    assert abs(sq[i, j].item() - want) <= 1e-6 * want
  sel = bm.gars.krum_selection(rows, f)
  # empire Byzantine rows sit at -0.1*mean: they are the closest to everyone and get selected first,
  # tied scores resolved by index (the reference's stable sort)
  assert sel[:f] == list(range(h, n))
  # the average of the selected rows equals torch's sequential sum on the same GPU
  want = sum(rows[i] for i in sel)                                  # same order, same fp32 adds
  assert close(bm.krum(rows, f), want / len(sel), 2e-7)             # torch-GPU divides by reciprocal
  order = bm.gars.bulyan_ranking(rows, f)
  assert sorted(order) == list(range(n))
  out = bm.bulyan(rows, f)
  assert bool(torch.isfinite(out).all())
  top = torch.stack([rows[i] for i in order[:n - f - 2]])
  assert bool((out <= top.max(dim=0).values).all() and (out >= top.min(dim=0).values).all())


def test_aggregation_step_matches_oracle_simulation(bm):
  """Three steps of the attack.py:800-878 mirror (worker momentum, empire attack, Krum, study
  statistics with past gradients) against the same loop written with the oracle."""
  from byzantinemomentum_amd.step import AggregationStep
  n, f, d, mu, damp, factor = 11, 2, 30011, 0.9, 0.9, 1.1
  h = n - f
  step = AggregationStep(n, f, f, gar="krum", momentum=mu, dampening=damp, attack_factor=factor, nb_past=3)
  gen = torch.Generator().manual_seed(123)
  bufs = [torch.zeros(d) for _ in range(h)]
  pasts = []
  for it in range(3):
    base = 0.2 * torch.randn(d, generator=gen)
    sampled = [base + (0.5 + 0.1 * i) * torch.randn(d, generator=gen) for i in range(h)]
    defense = step.run([g.to(DEV) for g in sampled])
    got = step.floats()
    # oracle loop
    O.worker_momentum(bufs, sampled, mu, damp)
    avg = torch.stack(bufs).mean(dim=0)
    att = avg.neg()
    att.mul_(factor)
    byz = avg.add(att)
    grads = list(bufs) + [byz] * f
    want_def = O.krum(grads, f)
    want = O.study_block(sampled, bufs, grads[h:], want_def, pasts, mu, "f64")
    scale = float(torch.stack(bufs).abs().max())
    assert close(defense, want_def, 2e-6, scale)
    for key in ("sampled_norm_avg", "honest_norm_avg", "attack_norm_avg", "defense_norm_avg", "sampled_norm_dev",
                "honest_norm_dev", "sampled_norm_max", "honest_norm_max", "attack_norm_max", "defense_norm_max"):
      assert abs(got[key] - want[key]) <= 1e-5 * max(abs(want[key]), 1e-3), (it, key, got[key], want[key])
    assert abs(got["attack_norm_dev"] - want["attack_norm_dev"]) <= 1e-5 * want["attack_norm_avg"]
    for key in ("cosin_splhon", "cosin_splatt", "cosin_spldef", "cosin_honatt", "cosin_hondef", "cosin_attdef",
                "cosin_sampled"):
      assert (math.isnan(got[key]) and math.isnan(want[key])) or abs(got[key] - want[key]) <= 1e-5, (it, key)
    if it > 0:
      assert abs(got["curv_sampled"] - want["curv_sampled"]) <= 1e-5 * max(abs(want["curv_sampled"]), 1.0)
    pasts.insert(0, (want["sampled_grad_avg"], want["sampled_norm_avg"]))
    pasts = pasts[:3]


def test_sharded_aggregator_hip_backend_single_rank(bm):
  """World size 1: the sharded front end must give exactly the plain rules and issue no collective."""
  from byzantinemomentum_amd.sharded import ShardedAggregator
  rows, h = O.make_stack("hetero", 25, 5, 40007, seed=17)
  dev = to_dev(rows)
  agg = ShardedAggregator()
  assert agg.world_size == 1 and not agg.collective
  assert torch.equal(agg.median(dev), bm.median(dev))
  assert torch.equal(agg.trmean(dev, 5), bm.trmean(dev, 5))
  assert torch.equal(agg.krum(dev, 5), bm.krum(dev, 5))
  assert torch.equal(agg.bulyan(dev, 5), bm.bulyan(dev, 5))
  assert torch.equal(agg.aksel(dev, 5), bm.aksel(dev, 5))
  avg, norm, devi, mx = agg.compute_avg_dev_max(dev[:h])
  want = bm.compute_avg_dev_max(dev[:h])
  assert torch.equal(avg, want[0]) and (norm, devi, mx) == want[1:]


def test_influence_hooks_match_reference_semantics(bm):
  """`native` influence functions: fraction of the selected gradients that ARE attack tensors
  (aggregators/krum.py:126-150, brute.py:118-140, aksel.py:83-105), from the selected indices."""
  import native
  rows, h = O.make_stack("hetero", 11, 2, 5003, seed=31)
  dev = to_dev(rows)
  honests, attacks = dev[:h], dev[h:]
  order, _ = O.krum_order(rows, 2)
  for m in (None, 1, 3):
    mm = 11 - 2 - 2 if m is None else m
    want = sum(1 for i in order[:mm] if i >= h) / mm
    assert native._influence_krum(honests, attacks, 2, m) == want
  sel = O.brute_selection(rows, 2)
  assert native._influence_brute(honests, attacks, 2) == sum(1 for i in sel if i >= h) / len(sel)
  ao, _ = O.aksel_order(rows)
  assert native._influence_aksel(honests, attacks, 2) == sum(1 for i in ao[:6] if i >= h) / 6
  assert native._influence_aksel(honests, attacks, 2, mode="n-f") == sum(1 for i in ao[:9] if i >= h) / 9
  co, _ = O.cge_order(rows)
  assert native._influence_cge(honests, attacks, 2) == sum(1 for i in co[:9] if i >= h) / 9
  # the ranking cache must not survive an in-place change of a gradient
  before = bm.gars.krum_selection(dev, 2)
  dev[0].mul_(50.0)
  after = bm.gars.krum_selection(dev, 2)
  assert 0 in before and 0 not in after


@pytest.mark.parametrize("n,f,d", [(25, 5, 300003), (7, 1, 262145), (28, 6, 270000), (4, 1, 1048577), (25, 5, 1310723),
                                   (13, 3, 2100001), (1, 0, 1048583)])
def test_long_columns_with_nan_and_inf(bm, n, f, d):
  """Long columns with NaN / inf sprinkled in and a d % 4 tail (several grid-stride trips per lane).  From
  2^20 coordinates on, BM_COL_BURST=1 sends median / trmean through the burst form of the column kernel with a
  full and a partial iteration (test_colwise_burst_form_at_short_lengths re-runs this test that way)."""
  gen = torch.Generator().manual_seed(n * 1000 + f)
  rows = [torch.randn(d, generator=gen) for _ in range(n)]
  rows[0][::1001] = math.nan
  rows[n - 1][5::7777] = math.inf
  dev = [r.to(DEV) for r in rows]
  assert same_bits(bm.median(dev), O.median(rows))
  got, want = bm.trmean(dev, f).cpu(), O.trmean(rows, f)
  assert bool((torch.isnan(got) == torch.isnan(want)).all())
  fin = torch.isfinite(want)
  assert bool(((got[fin] - want[fin]).abs() <= 1e-6 * 5).all())
  assert bool((got[~fin & ~torch.isnan(want)] == want[~fin & ~torch.isnan(want)]).all())


def test_colwise_burst_form_at_short_lengths():
  """The burst form of median / trmean (one workgroup per CU, results staged in LDS and written in bursts) is used
  from 8 iterations per CU on (d >= 8.4 M: the full-size tests); BM_COL_BURST=1 lowers that to one iteration so that
  the NaN / inf / tail cases above run through it too.  The knob is read once per process: subprocess."""
  env = dict(os.environ, BM_COL_BURST="1", PYTHONPATH=ROOT)
  out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x",
                        "-m", "gpu", "-k", "test_long_columns_with_nan_and_inf or test_full_size_colwise_properties"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
  assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])


def test_fused_attack_vectors(bm):
  """Byzantine vectors produced in the honest-stack statistics pass (attacks/identical.py:63-86)."""
  rows, h = O.make_stack("iid", 20, 0, 40003, seed=5)
  dev = [r.to(DEV) for r in rows]
  stck = torch.stack(rows)
  avg = stck.mean(dim=0)
  _, _, emp = bm.stats.stack_stats_async(dev, scale=1.1, attack="empire")
  want = avg + 1.1 * avg.neg()
  assert close(emp, want, 2e-6, float(avg.abs().max()))
  _, _, lit = bm.stats.stack_stats_async(dev, scale=-1.5, attack="little")
  want = avg + (-1.5) * stck.var(dim=0).sqrt_()
  assert close(lit, want, 2e-6, float(want.abs().max()))


def test_direct_difference_pairwise_mode():
  """BM_PAIR_MODE=1 (the VALU direct-difference kernel) is read once per process: run it in a
  subprocess and compare squared distances with float64, including ties of aliased rows."""
  import os
  import subprocess
  import sys
  code = (
    "import torch, math\n"
    "import byzantinemomentum_amd as bm\n"
    "from oracle import gar_oracle as O\n"
    "for n, f, d in ((13, 3, 4099), (51, 12, 20001)):\n"
    "  rows, h = O.make_stack('hetero', n, f, d, seed=4)\n"
    "  seen = {}\n"
    "  dev = [seen.setdefault(id(g), g.to('cuda:0')) for g in rows]\n"
    "  sq = bm.gars.pairwise_sqdist(dev).cpu()\n"
    "  want = torch.from_numpy(O.pairwise_distances(rows, 'f64')) ** 2\n"
    "  assert ((sq - want).abs() <= 1e-6 * want.max()).all()\n"
    "  assert torch.equal(sq, sq.T) and sq[h, h + 1].item() == 0.0 and torch.equal(sq[h, :h], sq[n - 1, :h])\n"
    "  assert bm.gars.krum_selection(dev, f) == O.krum_order(rows, f, 'f64')[0][:n - f - 2]\n"
    "print('direct ok')\n")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, BM_PAIR_MODE="1", PYTHONPATH=root)
  out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root)
  assert out.returncode == 0 and "direct ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("n,f", [(64, 15), (63, 15), (64, 1), (48, 11), (49, 11), (33, 7), (17, 3), (16, 3)])
def test_largest_row_counts_distance_rules(bm, n, f):
  """Upper end of the supported worker counts (BM_MAX_ROWS = 64) and the 16-row block boundaries of
  the Gram kernel: Krum, Bulyan (generic LDS kernel with > 64 KB of dynamic LDS at n = 64) and Aksel."""
  d = 6007
  rows, h = O.make_stack("hetero", n, f, d, seed=n * 31 + f)
  dev = to_dev(rows)
  m = n - f - 2
  assert bm.gars.krum_selection(dev, f) == O.krum_order(rows, f, "f64")[0][:m]
  assert torch.equal(bm.krum(dev, f).cpu(), O.krum(rows, f))
  sq = bm.gars.pairwise_sqdist(dev).cpu()
  want = torch.from_numpy(O.pairwise_distances(rows, "f64")) ** 2
  assert close(sq, want, 1e-6, float(want.max()) * 1e-3)
  if n >= 4 * f + 3:
    assert bm.gars.bulyan_ranking(dev, f) == O.bulyan_order(rows, f, None, "f64")[0]
    scale = float(torch.stack(rows[:h]).abs().max())
    assert close(bm.bulyan(dev, f), O.bulyan(rows, f), 2e-6, scale)
  assert bm.gars.aksel_selection(dev, f) == O.aksel_order(rows, "f64")[0][:(n + 1) // 2]
