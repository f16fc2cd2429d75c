"""Round-3 GPU parity: what the round-2 review found unpinned.

* the STEADY STATE of the study block (attack.py:861-868): the deque of past sampled averages full and its
  oldest entry leaving, over nb_past + 5 iterations, for the single-call step (bm_step_worker), the
  kernel-by-kernel sequence and the other momentum placements, against the independent loop of
  oracle/step_oracle.py; and at C5 size against fp64 torch reductions of the deque itself;
* STRUCTURED stacks through the default distance path (krum.py:41-63): rows with few distinct values (constant
  + small noise, sign, int8-quantised, top-1 % sparsified with exact zeros) at the lengths where the two-plane
  split of gram_bf16.hip is in use, every squared distance within 1e-5 of fp64 direct differences;
* Aksel (aksel.py:35-64) and CGE (cge.py:28-57) at C2 size against fp64 torch on the same GPU.
Needs an MI355X: `pytest -m gpu`.
"""

import math
import os

import numpy as np
import pytest
import torch

from oracle import gar_oracle as O
from tests.golden_io import same_bits
from tests.test_gpu_parity_r2 import DEV, D_RESNET18, D_WRN, sqdist_f64_on_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bm():
  import byzantinemomentum_amd
  byzantinemomentum_amd._lib.load()
  return byzantinemomentum_amd


# ---------------------------------------------------------------------------- #
# Steady state of the curvature recurrence

STEADY = [
  dict(gar="krum", momentum_at="worker", single_call=True, clip=None, nb_past=3),
  dict(gar="krum", momentum_at="worker", single_call=False, clip=None, nb_past=3),
  dict(gar="median", momentum_at="worker", single_call=True, clip=100.0, nb_past=2),
  dict(gar="bulyan", momentum_at="worker", single_call=True, clip=None, nb_past=1),
  dict(gar="trmean", momentum_at="server", single_call=False, clip=None, nb_past=3),
  dict(gar="median", momentum_at="update", single_call=False, clip=95.0, nb_past=4),
]


@pytest.mark.parametrize("cfg", STEADY, ids=lambda c: f"{c['gar']}-{c['momentum_at']}-{'one' if c['single_call'] else 'seq'}"
                                                      f"-past{c['nb_past']}-clip{c['clip']}")
def test_step_steady_state_against_reference_loop(bm, cfg):
  """nb_past + 5 iterations: the deque is full from iteration nb_past on, and from nb_past + 1 on the
  curvature seen by floats() depends on the entry that LEFT being removed with the right weight
  (step.py `oldest_weight`, step_call.hip).  The sampled averages share a persistent component, so every
  term of mu * sum_i mu^i <s, past_i> is large and of the same sign: a wrong weight on one of them is a
  relative error of order 1/nb_past, not something a tolerance hides."""
  from byzantinemomentum_amd.step import AggregationStep
  from tests.step_reference import ReferenceLoop, assert_floats_close
  n, f, d, mu = 11, 2, 30011, 0.9
  h = n - f
  P = cfg["nb_past"]
  step = AggregationStep(n, f, f, gar=cfg["gar"], momentum=mu, dampening=0.9, momentum_at=cfg["momentum_at"],
                         attack="empire", attack_factor=1.1, nb_past=P, gradient_clip=cfg["clip"],
                         single_call=cfg["single_call"])
  assert step.single_call == cfg["single_call"]
  ref = ReferenceLoop(n, f, f, cfg["gar"], cfg["momentum_at"], mu, 0.9, "empire", 1.1, cfg["clip"], P)
  gen = torch.Generator().manual_seed(2025)
  origin = torch.randn(d, generator=gen)
  params = origin.clone()
  drift = 0.3 * torch.randn(d, generator=gen)   # persistent across steps: <s_t, s_u> ~ 0.09 d for every t, u
  curvs = []
  for it in range(P + 5):
    base = drift + 0.1 * torch.randn(d, generator=gen)
    sampled = [base + (0.5 + 0.1 * i) * torch.randn(d, generator=gen) for i in range(h)]
    want_def, want_upd, want = ref.step(sampled, params, origin)
    got_def = step.run([g.to(DEV) for g in sampled], params.to(DEV), origin.to(DEV))
    scale = float(torch.stack(sampled).abs().max())
    assert float((got_def.cpu() - want_def).abs().max()) <= 4e-6 * scale, (cfg, it)
    assert float((step.update_gradient().cpu() - want_upd).abs().max()) <= 4e-6 * scale, (cfg, it)
    if it != P:  # one skipped read in the middle: the recurrence must advance whether or not floats() is called
      got = step.floats()
      assert_floats_close(got, want, tag=(cfg, it), tol=1e-5)
      if it >= 1:
        assert abs(got["curv_sampled"] - want["curv_sampled"]) <= 1e-5 * abs(want["curv_sampled"]), (cfg, it)
        curvs.append(want["curv_sampled"])
    params = params - 0.05 * want_upd
  # the test is meaningful: the curvature is dominated by the persistent component (~ mu * sum mu^i * 0.09 d)
  # (clipping scales the sampled gradients down by up to ~0.6, hence the 0.2)
  full = mu * sum(mu ** i for i in range(P)) * 0.09 * d
  assert curvs[-1] > (0.5 if cfg["clip"] is None else 0.2) * full, (curvs[-1], full)


def test_full_size_c5_steady_state_curvature(bm):
  """BASELINE config 5 on one GPU (d = 36 546 980, n = 25, f = 5, worker momentum, the single-call step) with
  nb_past = 3 for 7 steps: cosin_sampled and curv_sampled against fp64 dot products with the deque of past
  sampled averages kept by the TEST (attack.py:861-868, computed term by term, no recurrence), momentum buffers
  against torch's own mul_/add_."""
  from byzantinemomentum_amd.step import AggregationStep
  n, f, d, mu, damp, P = 25, 5, D_WRN, 0.99, 0.99, 3
  h = n - f
  gen = torch.Generator(device=DEV).manual_seed(78)
  drift = 0.1 * torch.randn(d, device=DEV, generator=gen)
  step = AggregationStep(n, f, f, gar="median", momentum=mu, dampening=damp, attack_factor=1.1, nb_past=P)
  assert step.single_call
  ref_bufs = [torch.zeros(d, device=DEV) for _ in range(h)]
  pasts = []  # newest first
  sigmas = torch.linspace(0.5, 1.5, h).tolist()
  for it in range(P + 4):
    sampled = [drift + s * torch.randn(d, device=DEV, generator=gen) for s in sigmas]
    defense = step.run(sampled)
    got = step.floats()
    for b, g in zip(ref_bufs, sampled):
      b.mul_(mu).add_(g, alpha=1.0 - damp)
    if it in (0, P + 3):
      for a, b in zip(step.buffers, ref_bufs):
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    if it == 0:  # the rule rode along with the first pass (fused kernel, burst form): it must be the median of
      # the step's own buffers and five copies of its Byzantine vector, bit for bit
      h_avg = step.buffers[0].clone()
      for bb in step.buffers[1:]:
        h_avg.add_(bb)
      h_avg.div_(torch.full_like(h_avg, float(h)))
      byz = h_avg + 1.1 * (-h_avg)
      want = torch.stack(list(step.buffers) + [byz] * f).median(dim=0).values
      assert torch.equal(defense, want)
      del want, h_avg, byz
    s_avg = sampled[0].clone()
    for r in sampled[1:]:
      s_avg.add_(r)
    s_avg.div_(h)
    s64 = s_avg.double()
    if pasts:
      want_cos = float(torch.dot(s64, pasts[0])) / math.sqrt(float(s64.pow(2).sum())) / \
          math.sqrt(float(pasts[0].pow(2).sum()))
      assert abs(got["cosin_sampled"] - want_cos) <= 1e-5, (it, got["cosin_sampled"], want_cos)
      want_curv = mu * sum(mu ** i * float(torch.dot(s64, p)) for i, p in enumerate(pasts))
      assert abs(got["curv_sampled"] - want_curv) <= 1e-5 * abs(want_curv), (it, got["curv_sampled"], want_curv)
    else:
      assert math.isnan(got["cosin_sampled"]) and math.isnan(got["curv_sampled"])
    pasts.insert(0, s64)
    del pasts[P:]
    del sampled, s_avg


# ---------------------------------------------------------------------------- #
# Structured stacks through the default distance path

def structured_stack(kind, n, f, d, seed):
  """Honest rows of few distinct values / low entropy, f aliased Byzantine rows (= -0.1 mean of the honest ones,
  the "empire" vector).  Generated on the GPU; the fp32 values ARE the inputs (quantisation happens before)."""
  gen = torch.Generator(device=DEV).manual_seed(seed)
  h = n - f
  randn = lambda: torch.randn(d, device=DEV, generator=gen)  # noqa: E731
  wts = torch.linspace(0.5, 2.0, h).tolist()
  if kind == "const+noise":          # w_i * 1 + 1e-3 randn
    honest = [w + 1e-3 * randn() for w in wts]
  elif kind == "const_close_pair":   # two rows 7 % apart (a pair just above the accuracy gate), the rest spread out
    wts[1] = wts[0] * 1.07
    honest = [w + 1e-4 * randn() for w in wts]
  elif kind == "const":              # exactly constant rows
    honest = [torch.full((d,), w, device=DEV) for w in wts]
  elif kind == "sign":               # 1-bit gradients: +-s_i
    honest = [(0.0123 * w) * torch.sign(randn()) for w in wts]
  elif kind == "int8":               # symmetric 8-bit quantisation of mu + sigma_i randn
    mu = 0.1 * randn()
    honest = []
    for w in wts:
      g = mu + w * randn()
      scale = float(g.abs().max()) / 127.0
      honest.append(torch.round(g / scale) * scale)
  elif kind == "top1%":              # keep the largest 1 % of the entries, exact zeros elsewhere
    mu = 0.1 * randn()
    honest = []
    for w in wts:
      g = mu + w * randn()
      k = d - d // 100
      thr = float(torch.kthvalue(g.abs(), k).values)
      honest.append(torch.where(g.abs() >= thr, g, torch.zeros_like(g)))
  elif kind == "ternary":            # {-s, 0, +s}
    honest = []
    for w in wts:
      g = randn()
      honest.append((0.01 * w) * (torch.sign(g) * (g.abs() > 0.7)))
  else:
    raise ValueError(kind)
  acc = torch.zeros(d, dtype=torch.float32, device=DEV)
  for g in honest:
    acc += g
  byz = acc.div_(h).mul_(-0.1)
  return honest + [byz] * f, h


STRUCTURED = ["const+noise", "const_close_pair", "const", "sign", "int8", "top1%", "ternary"]


@pytest.mark.parametrize("d", [(1 << 20) + 192, 2500003, D_RESNET18], ids=lambda d: f"d{d}")
@pytest.mark.parametrize("kind", STRUCTURED)
def test_structured_stacks_default_distance_path(bm, kind, d):
  """Every squared distance within 1e-5 (relative to itself) of fp64 direct differences on the same GPU, and
  the Multi-Krum / Bulyan selections equal to the oracle's ranking logic on those fp64 distances wherever the
  fp64 scores themselves separate the sets by more than the tolerance."""
  n, f = (25, 5) if d != 2500003 else (51, 12)
  rows, h = structured_stack(kind, n, f, d, seed=99)
  want = sqdist_f64_on_gpu(rows)
  sq = bm.gars.pairwise_sqdist(rows).cpu().numpy()
  assert np.array_equal(sq, sq.T) and not sq.diagonal().any()
  pos = want > 0
  assert not sq[~pos].any(), "bitwise-equal rows must give an exact zero"
  rel = np.abs(sq - want)[pos] / want[pos]
  assert rel.max() <= 1e-5, f"{kind}, d={d}: worst relative error of a squared distance {rel.max():.2e}"
  for a in range(h + 1, n):  # aliased Byzantine rows: bitwise-equal rows of the matrix
    assert np.array_equal(sq[h, :h], sq[a, :h])
  m = n - f - 2
  scores = O.krum_scores(np.sqrt(want), f)
  order = O._stable_order(scores)
  srt = sorted(scores)
  if srt[m] - srt[m - 1] > 1e-5 * srt[m]:
    assert sorted(bm.gars.krum_selection(rows, f)) == sorted(order[:m]), kind
  theta = n - 2 * f - 2
  if theta >= 1:
    bscores = [O._sum_smallest([math.sqrt(want[i, j]) for j in range(n) if j != i], m) for i in range(n)]
    border = O._stable_order(bscores)
    bsrt = sorted(bscores)
    if bsrt[m] - bsrt[m - 1] > 1e-5 * bsrt[m]:
      assert sorted(bm.gars.bulyan_ranking(rows, f)[:m]) == sorted(border[:m]), kind


def test_structured_stack_sharded_plan_follows_total_length(bm):
  """A shard of 2^17 coordinates of a 2^20-coordinate job takes the precision plan of the whole job
  (bm_pairwise_sqdist_shard): the partial matrices of the 8 shards add up to the unsharded matrix within the
  fp64 rounding of the sum, on a stack where the two plans differ measurably (constant rows)."""
  n, f, d, P = 13, 3, 1 << 20, 8
  rows, h = structured_stack("const+noise", n, f, d, seed=5)
  whole = bm.gars.pairwise_sqdist(rows).cpu().numpy()
  per = d // P
  acc = np.zeros((n, n))
  for p in range(P):
    acc += bm.gars.pairwise_sqdist([r[p * per:(p + 1) * per] for r in rows], d_total=d).cpu().numpy()
  want = sqdist_f64_on_gpu(rows)
  pos = want > 0
  assert (np.abs(acc - want)[pos] / want[pos]).max() <= 1e-5
  assert (np.abs(acc - whole)[pos] / want[pos]).max() <= 1e-6


# ---------------------------------------------------------------------------- #
# Aksel and CGE at C2 size

def test_full_size_aksel_and_cge_against_fp64(bm):
  n, f, d = 25, 5, D_RESNET18
  gen = torch.Generator(device=DEV).manual_seed(404)
  mu = 0.1 * torch.randn(d, device=DEV, generator=gen)
  h = n - f
  honest = [mu + s * torch.randn(d, device=DEV, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()]
  acc = torch.zeros(d, device=DEV)
  for g in honest:
    acc += g
  rows = honest + [acc.div_(h).mul_(-0.1)] * f
  # Aksel (aksel.py:35-48): squared distances to the coordinate-wise (lower) median, fp64 on the same GPU
  med = torch.stack(rows).median(dim=0).values
  sq64 = [float((r.double() - med.double()).pow(2).sum()) for r in rows]
  got_sq = bm.gars.aksel_sqdist(rows)[:n].cpu().tolist()
  for a, b in zip(got_sq, sq64):
    assert abs(a - b) <= 1e-5 * b
  for a in range(h + 1, n):
    assert got_sq[a] == got_sq[h]  # aliased rows: exact ties
  order = O._stable_order(sq64)
  for mode, count in (("mid", (n + 1) // 2), ("n-f", n - f)):
    sel = bm.gars.aksel_selection(rows, f, mode)
    srt = sorted(sq64)
    assert srt[count] - srt[count - 1] > 1e-5 * srt[count]
    assert sorted(sel) == sorted(order[:count]), mode
    ref = torch.zeros(d, device=DEV)
    for i in sel:   # torch's own sequential fp32 sum in the order the rule used (aksel.py:64)
      ref = ref + rows[i]
    assert same_bits(bm.aksel(rows, f, mode), ref.cpu().div_(count)), mode
  # CGE (cge.py:28-57): rows by increasing norm, mean of the n - f smallest
  norms = [math.sqrt(float(r.double().pow(2).sum())) for r in rows]
  corder = O._stable_order(norms)
  keep = n - f
  sel = bm.gars.cge_selection(rows, f)[:keep].cpu().tolist()
  srt = sorted(norms)
  assert srt[keep] - srt[keep - 1] > 1e-5 * srt[keep]
  assert sorted(sel) == sorted(corder[:keep])
  byz = [i for i in sel if i >= h]
  assert byz == sorted(byz)  # tied rows in index order (stable)
  ref = rows[sel[0]].clone()
  for i in sel[1:]:
    ref.add_(rows[i])
  assert same_bits(bm.cge(rows, f), ref.cpu().div_(keep))


# ---------------------------------------------------------------------------- #
# The burst form of bm_momentum_stats (one workgroup per CU, barrier between the loads and the stores) is the default
# from 8 iterations per CU on (4.2 M coordinates: the C5-size tests run it); here at every length

def test_momentum_stats_burst_form_at_short_lengths():
  import subprocess
  import sys
  from tests.test_gpu_parity_r2 import ROOT
  env = dict(os.environ, BM_STEP_BURST="1", PYTHONPATH=ROOT)
  out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity_r2.py"),
                        os.path.join(ROOT, "tests", "test_gpu_parity_r3.py"), "-q", "-x",
                        "-m", "gpu", "-k", "test_momentum_stats_kernel_tiers or test_step_all_placements or "
                        "test_single_call_step_equals_python_sequence or test_first_pass_with_the_rule_riding_along"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
  assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])


def test_study_stats_burst_form_at_short_lengths():
  """The burst form of bm_study_stats (one workgroup of 1024 lanes per CU, C staged in LDS and written in chip-wide
  bursts, fp32 chains folded into fp64 at every burst) at lengths of one to a few iterations per CU, ragged tails
  included: the same tests as the plain form, in a process that reads BM_STUDY_BURST=1."""
  import subprocess
  import sys
  from tests.test_gpu_parity_r2 import ROOT
  env = dict(os.environ, BM_STUDY_BURST="1", PYTHONPATH=ROOT)
  out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity_r3.py"), "-q", "-x",
                        "-m", "gpu", "-k", "test_study_stats_against_fp64"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
  assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])


def test_study_stats_against_fp64(bm):
  """bm_study_stats alone, every curvature mode, with and without attack / l2, odd lengths and unaligned views:
  every slot against fp64 torch, C against the same fp32 operations."""
  gen = torch.Generator(device=DEV).manual_seed(11)
  # (the two long ones reach the burst form when BM_STUDY_BURST=1 lowers its threshold to one iteration per CU:
  #  test_study_stats_burst_form_at_short_lengths; by default it starts at 8.4 M coordinates, the C5-size tests)
  for d, off in ((100003, 0), (4099, 1), (64, 0), (3, 0), (262147, 2), (1200007, 0), (2400002, 0)):
    def vec():
      return torch.randn(d + off, device=DEV, generator=gen)[off:]
    s, h, df, byz, past, old, par, org = (vec() for _ in range(8))
    for f_real in (0, 1, 5):
      for mode in (0, 1, 2, 3):
        for l2 in (False, True):
          curv0 = vec()
          curv = curv0.clone()
          mu, w = 0.9, -(0.9 ** 4)
          out = bm.stats.study_stats(s, h, df, byz if f_real else None, f_real, past_newest=past if mode >= 2 else None,
                                     curv=curv if mode >= 1 else None, past_oldest=old if mode == 3 else None,
                                     curv_mode=mode, mu=mu, oldest_weight=w, params=par if l2 else None,
                                     origin=org if l2 else None).tolist()
          core = [s, h, df]
          if f_real:
            a = byz.clone()
            for _ in range(f_real - 1):
              a = a + byz
            a = a / torch.full_like(a, float(f_real))  # true division like the kernel and torch-CPU (a Python scalar becomes x * (1/f) on the GPU)
            core.append(a)
          c64 = [c.double() for c in core]
          for i in range(4):
            for j in range(4):
              want = float(torch.dot(c64[i], c64[j])) if i < len(core) and j < len(core) else 0.0
              scale = math.sqrt(float(c64[i].pow(2).sum()) * float(c64[j].pow(2).sum())) if want else 1.0
              assert abs(out[4 * i + j] - want) <= 1e-6 * scale, (d, f_real, mode, i, j)
          if f_real:
            assert abs(out[18] - float(c64[3].pow(2).sum())) <= 1e-6 * out[18]
            dev = f_real * float((byz.double() - c64[3]).pow(2).sum())
            assert abs(out[19] - dev) <= 1e-5 * dev + 1e-30, (out[19], dev)
            assert out[20] == float(core[3].abs().max())
          else:
            assert out[18] == out[19] == out[20] == 0.0
          assert out[21] == float(df.abs().max())
          if mode >= 2:
            for slot, other in ((16, past), (17, curv0)):
              want = float(torch.dot(c64[0], other.double()))
              assert abs(out[slot] - want) <= 1e-6 * math.sqrt(float(c64[0].pow(2).sum()) * float(other.double().pow(2).sum()))
          else:
            assert out[16] == out[17] == 0.0
          want_l2 = float((par.double() - org.double()).pow(2).sum()) if l2 else 0.0
          assert abs(out[22] - want_l2) <= 1e-6 * want_l2
          if mode == 0:
            assert torch.equal(curv, curv0)
          elif mode == 1:
            assert torch.equal(curv, s)
          else:
            t = curv0 if mode == 2 else torch.addcmul(curv0, torch.full_like(old, w), old)  # fma(w, oldest, C)
            want_c = s + mu * t
            assert float((curv - want_c).abs().max()) <= 2e-7 * float(want_c.abs().max()), (d, mode)
  # NaN in the defense vector / the Byzantine vector propagates to the maxima like torch's abs().max()
  s, h, df, byz = (torch.randn(1000, device=DEV, generator=gen) for _ in range(4))
  df[17] = float("nan")
  byz[3] = float("nan")
  out = bm.stats.study_stats(s, h, df, byz, 2).tolist()
  assert math.isnan(out[20]) and math.isnan(out[21])


def test_row_sqnorms_against_fp64(bm):
  """bm_row_sqnorms (cge.py:28-38, the clipping norms of attack.py:791-794): every row count tier, odd lengths,
  unaligned views, empty input; non-finite rows give non-finite norms (CGE ranks them last)."""
  gen = torch.Generator(device=DEV).manual_seed(21)
  for k, d, off in ((1, 5, 0), (3, 100003, 0), (25, 40009, 1), (64, 1027, 2), (7, 300000, 0), (4, 0, 0)):
    rows = [(1.0 + i) * torch.randn(d + off, device=DEV, generator=gen)[off:] for i in range(k)]
    got = bm.stats.row_sqnorms(rows).tolist()
    for g, r in zip(got, rows):
      want = float(r.double().pow(2).sum())
      assert abs(g - want) <= 1e-6 * want, (k, d, g, want)
  rows = [torch.randn(1000, device=DEV, generator=gen) for _ in range(5)]
  rows[2][7] = float("inf")
  rows[4][0] = float("nan")
  got = bm.stats.row_sqnorms(rows).tolist()
  assert math.isinf(got[2]) and math.isnan(got[4]) and all(math.isfinite(got[i]) for i in (0, 1, 3))
  assert bm.gars.cge_selection(rows, 2)[:3].cpu().tolist() == sorted(range(5), key=lambda i: got[i] if math.isfinite(got[i]) else math.inf)[:3]


def test_rows_from_the_package_allocator(bm):
  """layout.alloc_rows: rows of one allocation at a skewed stride are ordinary inputs (same bits out of every rule as
  separately allocated tensors) and 256-byte aligned."""
  from byzantinemomentum_amd.layout import alloc_rows, ROW_SKEW_BYTES
  n, f = 11, 2
  for d in (300007, 1 << 20, 77):
    rows, h = O.make_stack("hetero", n, f, d, seed=3)
    sep = [r.to(DEV) for r in rows]
    slab = alloc_rows(n, d, DEV)
    for a, b in zip(slab, sep):
      a.copy_(b)
    assert all(t.data_ptr() % 256 == 0 and t.is_contiguous() for t in slab)
    assert (slab[1].data_ptr() - slab[0].data_ptr()) % 256 == 0 and slab[0].untyped_storage().data_ptr() == slab[-1].untyped_storage().data_ptr()
    if d * 4 >= 1 << 20:
      assert (slab[1].data_ptr() - slab[0].data_ptr()) % (2 << 20) == ROW_SKEW_BYTES
    for rule in (bm.median, lambda g: bm.trmean(g, f), lambda g: bm.krum(g, f), lambda g: bm.bulyan(g, f),
                 lambda g: bm.aksel(g, f), lambda g: bm.cge(g, f)):
      assert torch.equal(rule(slab), rule(sep))
    assert torch.equal(bm.gars.pairwise_sqdist(slab), bm.gars.pairwise_sqdist(sep))
  # successive allocations continue the skew sequence: the rows of a second stack do not line up with the first one's
  again = alloc_rows(n, 1 << 20, DEV)
  assert (again[0].data_ptr() - slab[0].data_ptr()) % (2 << 20) != 0


def test_first_pass_with_the_rule_riding_along(bm):
  """bm_momentum_stats_colwise against bm_momentum_stats + bm_colwise on clones: same bits in the buffers, the two
  averages, the Byzantine vector, the six statistics and the aggregated vector — for the shapes with a fused
  instance (ks = h = 20 with 1..6 Byzantine copies or 14 with 11; median / trmean / phocas / meamed, with and without
  clipping factors, NaN / inf columns, ragged tails) and for shapes that fall back to the two kernels (other row
  counts, unaligned views)."""
  gen = torch.Generator(device=DEV).manual_seed(17)
  cases = [(20, 20, 5, "median", 0, 30011, 0, False), (20, 20, 5, "trmean", 5, 30011, 0, True),
           (20, 20, 1, "trmean", 3, 4099, 0, False), (20, 20, 6, "median", 0, 1 << 20, 0, True),
           (20, 20, 3, "trmean", 0, 65, 0, False), (20, 20, 5, "median", 0, 40002, 1, False),   # unaligned: fallback
           (21, 20, 5, "median", 0, 10007, 0, False), (12, 12, 3, "trmean", 2, 10007, 0, False),
           (20, 20, 7, "median", 0, 5003, 0, False), (20, 20, 5, "phocas", 5, 20011, 0, False),
           (20, 20, 5, "meamed", 5, 20011, 0, True),
           (14, 14, 11, "median", 0, 30011, 0, False), (14, 14, 11, "trmean", 11, 40003, 0, True),   # n = 25, f = 11
           (14, 14, 11, "phocas", 11, 30011, 0, False), (14, 14, 11, "meamed", 11, 1 << 20, 0, True),
           (20, 20, 5, "phocas", 5, (1 << 20) + 3, 0, True),
           (14, 14, 10, "median", 0, 5003, 0, False)]                                                  # no instance: fallback
  for ks, h, nb, rule, f, d, off, clip in cases:
    sampled = [torch.randn(d + off, device=DEV, generator=gen)[off:] for _ in range(ks)]
    bufs = [torch.randn(d + off, device=DEV, generator=gen)[off:] for _ in range(h)]
    if d > 100:
      sampled[3][17] = float("nan")
      sampled[5][40] = float("inf")
      bufs[2][63] = float("-inf")
    factors = None
    if clip:
      factors = torch.ones(64, device=DEV)
      factors[1], factors[ks - 1] = 0.5, 0.25
    for attack, scale in (("empire", 1.1), ("little", -1.5)):
      b1 = [b.clone() for b in bufs]
      b2 = [b.clone() for b in bufs]
      s1, h1, z1, d1, o1 = bm.stats.momentum_stats_colwise(sampled, b1, 0.9, 0.1, factors, scale, attack, rule, f, nb)
      s2, h2, z2, o2 = bm.stats.momentum_stats(sampled, b2, 0.9, 0.1, factors, scale, attack)
      rows = b2 + [z2] * nb
      d2 = bm.median(rows) if rule == "median" else getattr(bm, rule)(rows, f)
      tag = (ks, h, nb, rule, f, d, off, clip, attack)
      for x, y in zip(b1, b2):
        assert same_bits(x, y), tag
      assert same_bits(s1, s2) and same_bits(h1, h2) and same_bits(z1, z2), tag
      assert same_bits(d1, d2), tag
      a, b = o1.tolist(), o2.tolist()
      assert all(x == y or (math.isnan(x) and math.isnan(y)) for x, y in zip(a, b)), (tag, a, b)


@pytest.mark.parametrize("d,h,nb", [(D_WRN, 20, 5), (4300800 + 2, 20, 5), (4300800 + 2, 14, 11), (4300800, 20, 2)])
def test_first_pass_with_the_distance_pass_riding_along(bm, d, h, nb):
  """bm_momentum_stats_sqdist at sizes where the fused kernel runs (ks = h = 20, 5 Byzantine copies; the second size
  has a two-column tail): buffers, averages, Byzantine vector and statistics with the bits of bm_momentum_stats, the
  25 x 25 squared distances within 1e-5 (relative to themselves) of fp64 direct differences on the same GPU, exact zeros
  among the Byzantine copies and bitwise-equal rows for them, selections of Krum and Bulyan equal to the stand-alone
  pass; with and without clipping factors, both attacks.  Shapes: n = 25 with f = 5 and with f = 11 (reproduce.py:181),
  and 20 + 2."""
  n = h + nb
  gen = torch.Generator(device=DEV).manual_seed(23)
  drift = 0.1 * torch.randn(d, device=DEV, generator=gen)
  sampled = [drift + s * torch.randn(d, device=DEV, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()]
  bufs = [0.3 * drift + 0.05 * torch.randn(d, device=DEV, generator=gen) for _ in range(h)]
  factors = torch.ones(64, device=DEV)
  factors[1], factors[h - 1] = 0.5, 0.25
  for attack, scale, clip in (("empire", 1.1, None), ("little", -1.5, factors)):
    b1 = [b.clone() for b in bufs]
    b2 = [b.clone() for b in bufs]
    s1, h1, z1, sq, o1 = bm.stats.momentum_stats_sqdist(sampled, b1, 0.99, 0.01, clip, scale, attack, nb)
    s2, h2, z2, o2 = bm.stats.momentum_stats(sampled, b2, 0.99, 0.01, clip, scale, attack)
    for x, y in zip(b1, b2):
      assert torch.equal(x, y)
    assert torch.equal(s1, s2) and torch.equal(h1, h2) and torch.equal(z1, z2)
    assert o1.tolist() == o2.tolist()
    rows = b2 + [z2] * nb
    got = sq.cpu().numpy()
    want = sqdist_f64_on_gpu(rows)
    assert np.array_equal(got, got.T) and not got.diagonal().any()
    pos = want > 0
    assert not got[~pos].any()
    rel = np.abs(got - want)[pos] / want[pos]
    assert rel.max() <= 1e-5, (attack, d, rel.max())
    for a in range(h + 1, n):
      assert np.array_equal(got[h, :h], got[a, :h])
    ref = bm.gars.pairwise_sqdist(rows).cpu().numpy()
    for f in (min(nb, (n - 3) // 2),):
      scores = O.krum_scores(np.sqrt(got), f)
      scores_ref = O.krum_scores(np.sqrt(ref), f)
      m = n - f - 2
      assert sorted(O._stable_order(scores)[:m]) == sorted(O._stable_order(scores_ref)[:m])
    del rows, want
  if d != D_WRN and (h, nb) == (20, 5):  # non-finite coordinates: the rows that hold them are at non-finite distance of everything (krum.py:46-47
    # turns that into +inf), every other distance is untouched — as in the stand-alone pass
    sampled[3][17] = float("nan")
    bufs[2][4000000] = float("inf")
    b1 = [b.clone() for b in bufs]
    s1, h1, z1, sq, o1 = bm.stats.momentum_stats_sqdist(sampled, b1, 0.99, 0.01, None, 1.1, "empire", nb)
    got = sq.cpu().numpy()
    rows = b1 + [z1] * nb
    ref = bm.gars.pairwise_sqdist(rows).cpu().numpy()
    bad_rows = [2, 3] + list(range(h, n))  # the Byzantine vector inherits both through the honest average
    ok = [i for i in range(n) if i not in bad_rows]
    for i in range(n):
      for j in range(n):
        if i == j:
          continue
        if i in ok and j in ok:
          assert math.isfinite(got[i, j]) and abs(got[i, j] - ref[i, j]) <= 1e-5 * ref[i, j], (i, j, got[i, j], ref[i, j])
        elif not (i >= h and j >= h):
          assert not math.isfinite(got[i, j]) or not math.isfinite(ref[i, j]) or abs(got[i, j] - ref[i, j]) <= 1e-5 * ref[i, j], (i, j)
    assert bm.gars.krum_selection(rows, 5) is not None


def bulyan_columns(rows, order, f, cols):
  """Pass 2 of Bulyan (bulyan.py:64-84) on the columns `cols` of `rows` in the reference's own fp32 arithmetic
  (sequential sums in rank order, true division, lower median, mean of the beta closest).  Returns (value, tie):
  tie = the beta-th smallest deviation from the median of `selected` EQUALS the (beta+1)-th, where
  `topk(..., sorted=False)` may keep either row."""
  n = len(rows)
  m, theta = n - f - 2, n - 2 * f - 2
  beta = theta - 2 * f
  sub = [g[cols] for g in rows]
  sel = torch.stack([O._seq_sum_div([sub[r] for r in order[i:m]], m - i) for i in range(theta)])
  centre = sel.median(dim=0).values
  dev = (sel - centre).abs().sort(dim=0).values
  return O._closest_like_reference(sel, beta, centre), dev[beta - 1] == dev[beta]


def closest_columns(rows, keep, cols, centre="median", f=None):
  """The same for meamed / phocas (trmean.py:35-50): (value, tie) per column, tie = the keep-th and (keep+1)-th
  smallest |g - centre| are equal."""
  sub = [g[cols] for g in rows]
  x = torch.stack(sub)
  c = x.median(dim=0).values if centre == "median" else O.trmean(sub, f)
  dev = (x - c).abs().sort(dim=0).values
  return O._closest_like_reference(x, keep, c), dev[keep - 1] == dev[keep]


def explained(got, value, tie, scale, tol=4e-6):
  """Every column: the kernel's result IS the reference rule's on the same inputs, or the column is an exact tie."""
  return bool((((got - value).abs() <= tol * scale) | tie).all())


@pytest.mark.parametrize("gar,f", [("krum", 5), ("bulyan", 5), ("median", 5), ("krum", 11), ("trmean", 11)])
def test_step_with_the_rule_fed_from_the_first_pass(bm, gar, f):
  """n = 25, f = 5 or 11, d = 4 300 802 (the fused kernels run: 20 / 14 honest workers, long enough for the burst form, a
  two-column tail): the single-call step and the kernel-by-kernel sequence give the same bits, and both match the
  independent loop of oracle/step_oracle.py (oracle arithmetic on the CPU) over two steps (the second with a
  non-zero momentum buffer and a past average)."""
  from byzantinemomentum_amd.step import AggregationStep
  from tests.step_reference import ReferenceLoop, assert_floats_close
  n, d = 25, 4300802
  h = n - f
  kw = dict(gar=gar, momentum=0.9, dampening=0.9, attack="empire", attack_factor=1.1, nb_past=2)
  one = AggregationStep(n, f, f, single_call=True, **kw)
  seq = AggregationStep(n, f, f, single_call=False, **kw)
  ref = ReferenceLoop(n, f, f, gar, "worker", 0.9, 0.9, "empire", 1.1, None, 2)
  gen = torch.Generator().manual_seed(5)
  drift = 0.2 * torch.randn(d, generator=gen)
  for it in range(2):
    sampled = [drift + (0.5 + 0.05 * i) * torch.randn(d, generator=gen) for i in range(h)]
    want_def, want_upd, want = ref.step(sampled)
    dev = [g.to(DEV) for g in sampled]
    a = one.run([g.clone() for g in dev])
    b = seq.run([g.clone() for g in dev])
    assert torch.equal(a, b), (gar, it)
    for x, y in zip(one.buffers, seq.buffers):
      assert torch.equal(x, y)
    fa, fb = one.floats(), seq.floats()
    for key in fb:
      assert fa[key] == fb[key] or (math.isnan(fa[key]) and math.isnan(fb[key])), (gar, it, key)
    scale = float(torch.stack(sampled).abs().max())
    bad = ((a.cpu() - want_def).abs() > 4e-6 * scale).nonzero().flatten()
    # Bulyan's last step keeps the beta values closest to the median: with 4.3 M columns a few of them have an EXACT tie
    # at the window edge, where the reference's topk keeps either value and this kernel the upper window (documented
    # deviation).  Every column out of tolerance must BE such a tie, in the reference's own arithmetic.
    if gar == "bulyan" and len(bad):
      # the inputs of the two rules differ in their last bit (the Byzantine vector is formed by different sums), so a
      # column at a NEAR tie can flip too: ask the question on the kernel's OWN inputs — the reference's arithmetic on the
      # device's rows must give the kernel's value there, or the column is an exact tie of those rows
      mine = [b[bad].cpu() for b in one.buffers] + [one.last_byzantine[bad].cpu()] * f
      value, tie = bulyan_columns(mine, O.bulyan_order(ref.last_gradients, f)[0], f, torch.arange(len(bad)))
      assert len(bad) <= 20 and explained(a[bad].cpu(), value, tie, scale), (gar, it, bad.tolist())
    else:
      assert len(bad) == 0, (gar, it, bad[:10].tolist())
    assert_floats_close(fa, want, tag=(gar, it), tol=1e-5)


@pytest.mark.parametrize("gar,f", [("median", 5), ("krum", 5), ("meamed", 11), ("krum", 11)])
def test_update_placement_with_the_rule_fed_from_the_statistics_pass(bm, gar, f):
  """`--momentum-at update` (the reference's default), n = 25, d = 4 300 800: the rule (or its distance pass) rides along
  with the pass that forms the statistics and the Byzantine vector of the sampled stack; two steps against the independent
  loop of oracle/step_oracle.py, and the fused entry points against the stand-alone kernels."""
  from byzantinemomentum_amd.step import AggregationStep
  from tests.step_reference import ReferenceLoop, assert_floats_close
  n, d = 25, 4300800
  h = n - f
  step = AggregationStep(n, f, f, gar=gar, momentum=0.9, dampening=0.9, momentum_at="update", attack="little",
                         attack_factor=1.5, nb_past=2)
  ref = ReferenceLoop(n, f, f, gar, "update", 0.9, 0.9, "little", 1.5, None, 2)
  gen = torch.Generator().manual_seed(6)
  drift = 0.2 * torch.randn(d, generator=gen)
  for it in range(2):
    sampled = [drift + (0.5 + 0.05 * i) * torch.randn(d, generator=gen) for i in range(h)]
    want_def, want_upd, want = ref.step(sampled)
    dev = [g.to(DEV) for g in sampled]
    got = step.run(dev)
    scale = float(torch.stack(sampled).abs().max())
    # closest-to-centre rules: a few of the 4.3 M columns have an EXACT tie at the window edge, where the reference's
    # topk keeps either value and this kernel the upper window (documented deviation)
    # Every column out of tolerance must BE such a tie in the reference's own arithmetic (and the update, whose momentum
    # carries a disagreement of step 0 into step 1, may only differ where the defense of this or an earlier step did).
    bad = ((got.cpu() - want_def).abs() > 4e-6 * scale).nonzero().flatten()
    if gar == "meamed" and len(bad):
      # (on the kernel's OWN inputs: the Byzantine vector differs from the reference's in its last bit, a near tie flips)
      mine = [g[bad].cpu() for g in dev] + [step.last_byzantine[bad].cpu()] * f
      value, tie = closest_columns(mine, n - f, torch.arange(len(bad)))
      assert len(bad) <= 50 and explained(got[bad].cpu(), value, tie, scale), (gar, it, bad.tolist())
    else:
      assert len(bad) == 0, (gar, it, bad[:10].tolist())
    excused = bad if it == 0 else torch.unique(torch.cat([excused, bad]))
    bad_upd = ((step.update_gradient().cpu() - want_upd).abs() > 4e-6 * scale).nonzero().flatten()
    assert set(bad_upd.tolist()) <= set(excused.tolist()), (gar, it)
    assert_floats_close(step.floats(), want, tag=(gar, it), tol=1e-5)
  # the entry points against the stand-alone kernels, same rows
  if gar in ("median", "meamed"):
    avg, byz, defense, o6 = bm.stats.stack_stats_colwise(dev, 1.5, "little", gar, f, f)
    avg2, o3, byz2 = bm.stats.stack_stats_async(dev, scale=1.5, attack="little")
    assert torch.equal(avg, avg2) and torch.equal(byz, byz2)
    want_d = bm.median(dev + [byz2] * f) if gar == "median" else bm.meamed(dev + [byz2] * f, f)
    assert torch.equal(defense, want_d)
  else:
    avg, byz, sq, o6 = bm.stats.stack_stats_sqdist(dev, 1.5, "little", f)
    avg2, o3, byz2 = bm.stats.stack_stats_async(dev, scale=1.5, attack="little")
    assert torch.equal(avg, avg2) and torch.equal(byz, byz2)
    ref_sq = sqdist_f64_on_gpu(dev + [byz2] * f)
    got_sq = sq.cpu().numpy()
    pos = ref_sq > 0
    assert not got_sq[~pos].any() and (np.abs(got_sq - ref_sq)[pos] / ref_sq[pos]).max() <= 1e-5
  a, b = o6.tolist(), o3.tolist()
  assert a[:3] == a[3:] and all(abs(x - y) <= 1e-6 * abs(y) for x, y in zip(a[3:], b))


# ---------------------------------------------------------------------------- #
# Brute at the C3 shape (brute.py:32-80): 1.6e11 subsets, which the reference's loop cannot finish

@pytest.mark.parametrize("n,f,kind", [(51, 12, "hetero"), (25, 5, "hetero"), (25, 11, "tight")])
def test_brute_at_shapes_beyond_enumeration(bm, n, f, kind):
  """The selection on the device's distances is the one brute.py:47-68 would return on fp64 distances of the same
  rows (first subset in lexicographic order of strictly smallest diameter; the f Byzantine rows are ONE aliased
  tensor: exact zero distances and exact ties), decided by the oracle's checker without enumerating subsets — and,
  at n = 25, f = 5, equal to the enumeration itself; the average has the bits of torch's sequential sum."""
  from tests.test_gpu_parity_r2 import gpu_stack
  d = D_RESNET18 if n == 51 else 2500003
  rows, h = gpu_stack(kind, n, f, d, seed=77)
  dist = np.sqrt(sqdist_f64_on_gpu(rows))
  sel = bm.gars.brute_selection(rows, f)
  assert O.brute_selection_is_the_references(dist, f, sel), sel
  if (n, f) == (25, 5):
    assert sel == O.brute_selection_from_distances(dist, f)
  acc = torch.zeros(d, dtype=torch.float32, device=DEV)
  for i in sel:
    acc = acc + rows[i]
  assert same_bits(bm.brute(rows, f), acc.cpu().div_(n - f))


# ---------------------------------------------------------------------------- #
# HIP graphs of whole rules (the launch-bound regime of a rank of an 8-GPU job)

def test_graphed_calls_replay_the_rules(bm):
  """byzantinemomentum_amd.graphs.GraphedCall: a rule over a fixed set of row buffers recorded into a HIP graph gives,
  at every replay, the bits of the eager call on the CURRENT contents of those buffers — also after the contents
  changed, also with the ranking cache of gars.py warm (the recording must hold the distance pass, not a cached
  ranking), and for the single-call sharded rules with their all-reduce inside (one-rank RCCL group, forced
  collectives: what a rank of a multi-GPU job records)."""
  import socket
  import torch.distributed as dist
  from byzantinemomentum_amd.graphs import GraphedCall
  from byzantinemomentum_amd.sharded import ShardedAggregator
  n, f, d = 25, 5, 1396800  # one rank's shard of C4 at 8 GPUs
  gen = torch.Generator(device=DEV).manual_seed(12)
  mu = 0.1 * torch.randn(d, device=DEV, generator=gen)
  rows = [mu + s * torch.randn(d, device=DEV, generator=gen) for s in torch.linspace(0.5, 1.5, n).tolist()]
  calls = {"bulyan": lambda: bm.bulyan(rows, f), "krum": lambda: bm.krum(rows, f), "median": lambda: bm.median(rows),
           "trmean": lambda: bm.trmean(rows, f), "aksel": lambda: bm.aksel(rows, f)}
  graphs = {}
  for name, fn in calls.items():
    fn()  # (warm ranking cache)
    graphs[name] = GraphedCall(fn)
  for round_ in range(3):
    for name, fn in calls.items():
      want = fn().clone()
      got = graphs[name]()
      assert torch.equal(got, want), (name, round_)
    # other contents at the same addresses: the ranking changes (row 3 becomes an outlier, then row 7)
    rows[3 + 4 * round_].mul_(4.0)
    torch.cuda.synchronize()
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
  try:
    agg = ShardedAggregator(force_collectives=True)
    assert agg.native is not None and agg.single_call
    g_b = GraphedCall(lambda: agg.bulyan(rows, f))
    g_k = GraphedCall(lambda: agg.krum(rows, f))
    for round_ in range(2):
      assert torch.equal(g_b(), bm.bulyan(rows, f)) and torch.equal(g_k(), bm.krum(rows, f))
      rows[11].mul_(-3.0)
      torch.cuda.synchronize()
  finally:
    dist.destroy_process_group()


# ---------------------------------------------------------------------------- #
# The factor search against the median from two order statistics of the honest rows

@pytest.mark.parametrize("n,f,d,attack,negative", [(25, 5, 1000003, "empire", False), (25, 11, 262144, "little", True),
                                                   (11, 2, 8400000, "empire", False)])
def test_median_factor_search_from_two_order_statistics_on_gpu(bm, n, f, d, attack, negative):
  """attacks/identical.py:67-77 against the median: `auto` takes every candidate as the middle of (candidate, lo, hi)
  (step.py), `generic` runs the median over the n rows once per evaluation like the reference.  Same candidates and
  the same objective at each, bit for bit — unaligned length, the burst form of the column kernel (8.4 M
  coordinates), columns with ties and a NaN."""
  from byzantinemomentum_amd.step import AggregationStep
  h = n - f
  gen = torch.Generator(device=DEV).manual_seed(31)
  base = 0.2 * torch.randn(d, device=DEV, generator=gen)
  honests = [base + (0.5 + 0.05 * i) * torch.randn(d, device=DEV, generator=gen) for i in range(h)]
  for g in honests:
    g[::9] = g[::9].round()
  traces = {}
  for mode in ("auto", "generic"):
    step = AggregationStep(n, f, f, gar="median", momentum=0.9, dampening=0.9, momentum_at="update", attack=attack,
                           attack_factor=1.1, nb_past=0, attack_evals=10, attack_negative=negative, line_search=mode)
    out = step.run([g.clone() for g in honests])
    traces[mode] = (step.last_factor, list(step.last_search), out)
  (fa, sa, oa), (fg, sg, og) = traces["auto"], traces["generic"]
  assert fa == fg and sa == sg and len(sa) == 10 and max(y for _, y in sa) > 0
  assert torch.equal(oa, og)
  # a NaN in one honest row: the rule is NaN there whatever the candidate, both forms say so
  honests[3][17] = math.nan
  got = {}
  for mode in ("auto", "generic"):
    step = AggregationStep(n, f, f, gar="median", momentum_at="update", attack=attack, nb_past=0, attack_evals=3,
                           attack_negative=negative, line_search=mode)
    step.run([g.clone() for g in honests])
    got[mode] = step.last_search
  assert all(xa == xg and math.isnan(ya) and math.isnan(yg) for (xa, ya), (xg, yg) in zip(got["auto"], got["generic"]))


@pytest.mark.parametrize("n,f,d,attack,negative", [(25, 5, 1000003, "empire", False), (15, 3, 300000, "little", True)])
def test_bulyan_factor_search_ranks_from_scalars_on_gpu(bm, n, f, d, attack, negative):
  """attacks/identical.py:67-77 against Bulyan: `auto` ranks every candidate stack on the host from the inner products
  of ONE distance pass over honests + [avg, avg + dir] (bm_attack_ranking) and runs bm_bulyan_pass2 alone on the
  vectors; `generic` runs the distance pass, the rank kernel and pass 2 per evaluation.  Same candidates; the same
  ranking gives the same vectors, so the objective values agree (to rounding at most if two scores nearly tie)."""
  from byzantinemomentum_amd.step import AggregationStep
  h = n - f
  gen = torch.Generator(device=DEV).manual_seed(41)
  base = 0.2 * torch.randn(d, device=DEV, generator=gen)
  honests = [base + (0.5 + 0.05 * i) * torch.randn(d, device=DEV, generator=gen) for i in range(h)]
  traces = {}
  for mode in ("auto", "generic"):
    step = AggregationStep(n, f, f, gar="bulyan", momentum=0.9, dampening=0.9, momentum_at="update", attack=attack,
                           attack_factor=1.1, nb_past=0, attack_evals=10, attack_negative=negative, line_search=mode)
    out = step.run([g.clone() for g in honests])
    traces[mode] = (step.last_factor, list(step.last_search), out)
  (fa, sa, oa), (fg, sg, og) = traces["auto"], traces["generic"]
  assert len(sa) == len(sg) == 10 and max(y for _, y in sa) > 0
  for (xa, ya), (xg, yg) in zip(sa, sg):
    assert xa == xg and abs(ya - yg) <= 1e-5 * abs(yg) + 1e-12, (xa, ya, yg)
  assert fa == fg
  scale = float(torch.stack(honests).abs().max())
  assert float((oa - og).abs().max()) <= 4e-6 * scale
