"""Round-4 GPU parity: what the round-3 review found unpinned.

* the factor-search forms of the median and of Bulyan (step.py, attacks/identical.py:67-77) anchored to the
  independent loop of oracle/step_oracle.py at n = 25 (round 3 anchored them at n = 11 only);
* the Nesterov sequence of attack.py:757-783 (`nesterov_lookahead`) over several steps, both placements;
* the Brute rule with the subset search ON THE DEVICE (no host round trip): the selections of the host search,
  graph capture;
* the evaluate-only form of the factor search for the trimmed mean, phocas and meamed.
Needs an MI355X: `pytest -m gpu`.
"""

import math

import pytest
import torch

from oracle import gar_oracle as O
from tests.test_gpu_parity_r2 import DEV

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bm():
  import byzantinemomentum_amd
  byzantinemomentum_amd._lib.load()
  return byzantinemomentum_amd


# ---------------------------------------------------------------------------- #
# Search forms at n = 25 against the reference loop

SEARCH_N25 = [
  dict(gar="median", momentum_at="update", attack="empire", evals=16),
  dict(gar="median", momentum_at="worker", attack="little", evals=9, negative=True),
  dict(gar="bulyan", momentum_at="worker", attack="little", evals=10),
  dict(gar="bulyan", momentum_at="update", attack="empire", evals=16),
  dict(gar="krum", momentum_at="worker", attack="empire", evals=16),
  dict(gar="trmean", momentum_at="worker", attack="empire", evals=12),
  dict(gar="meamed", momentum_at="update", attack="little", evals=8),
  dict(gar="phocas", momentum_at="server", attack="empire", evals=8, negative=True),
]


@pytest.mark.parametrize("cfg", SEARCH_N25, ids=lambda c: f"{c['gar']}-{c['momentum_at']}-{c['attack']}-search{c['evals']}"
                                                          f"{'neg' if c.get('negative') else ''}")
def test_search_forms_at_n25_against_reference_loop(bm, cfg):
  """Same candidates in the same order, the same objective at each (hence the same decisions of
  tools/misc.py:468-514 and the same factor), the same aggregated gradient and study floats — n = 25, f = 5."""
  from byzantinemomentum_amd.step import AggregationStep
  from tests.step_reference import ReferenceLoop, assert_floats_close
  n, f, d = 25, 5, 50021
  h = n - f
  step = AggregationStep(n, f, f, gar=cfg["gar"], momentum=0.9, dampening=0.9, momentum_at=cfg["momentum_at"],
                         attack=cfg["attack"], nb_past=2, attack_evals=cfg["evals"],
                         attack_negative=cfg.get("negative", False))
  ref = ReferenceLoop(n, f, f, cfg["gar"], cfg["momentum_at"], 0.9, 0.9, cfg["attack"], 1.1, None, 2,
                      evals=cfg["evals"], negative=cfg.get("negative", False))
  gen = torch.Generator().manual_seed(2026)
  origin = torch.randn(d, generator=gen)
  params = origin.clone()
  for it in range(2):
    base = 0.2 * torch.randn(d, generator=gen)
    sampled = [base + (0.5 + 0.05 * i) * torch.randn(d, generator=gen) for i in range(h)]
    want_def, want_upd, want = ref.step(sampled, params, origin)
    got_def = step.run([g.to(DEV) for g in sampled], params.to(DEV), origin.to(DEV))
    got_search, want_search = step.last_search, ref.last_search
    assert len(got_search) == len(want_search) == cfg["evals"]
    floor = 1e-8 * want["honest_norm_dev"] ** 2 * h
    for (x, y), (xo, yo) in zip(got_search, want_search):
      assert x == xo and abs(y - yo) <= 2e-5 * abs(yo) + floor, (cfg, it, x, y, yo)
    assert step.last_factor == ref.last_factor, (cfg, it)
    scale = float(torch.stack(sampled).abs().max()) * max(1.0, abs(ref.last_factor))
    bad = ((got_def.cpu() - want_def).abs() > 4e-6 * scale).nonzero().flatten()
    if len(bad):  # closest-to-centre rules: on the kernel's own inputs, the reference's value or an exact tie
      from tests.test_gpu_parity_r3 import bulyan_columns, closest_columns, explained
      assert cfg["gar"] in ("bulyan", "meamed", "phocas") and len(bad) <= 3, (cfg, it, bad.tolist())
      # (honest rows: the device's momentum buffers; in the other placements the reference's rows, whose bits the
      #  device's share — sampled gradients as they are, or one fused multiply-add of them)
      hon = step.buffers if cfg["momentum_at"] == "worker" else ref.last_gradients[:h]
      mine = [b[bad].cpu() for b in hon] + [step.last_byzantine[bad].cpu()] * f
      cols = torch.arange(len(bad))
      if cfg["gar"] == "bulyan":
        value, tie = bulyan_columns(mine, O.bulyan_order(ref.last_gradients, f)[0], f, cols)
      else:
        value, tie = closest_columns(mine, n - f, cols, "median" if cfg["gar"] == "meamed" else "trmean", f)
      assert explained(got_def[bad].cpu(), value, tie, scale), (cfg, it, bad.tolist())
    assert_floats_close(step.floats(), want, tag=(cfg["gar"], it), tol=1e-5)
    params = params - 0.05 * want_upd


# ---------------------------------------------------------------------------- #
# Nesterov momentum (attack.py:757-783)

@pytest.mark.parametrize("momentum_at", ["worker", "update"])
def test_nesterov_sequence_against_the_reference_order_of_operations(bm, momentum_at):
  """attack.py:757-783 with `--momentum-nesterov`: before each worker's gradient the parameters move by
  -momentum * lr * (that worker's momentum buffer | the server momentum), the gradient is taken THERE, and the
  parameters are restored.  The gradient is a fixed function of the parameters here (g_i = a_i * theta + b_i), so a
  wrong shift, a shift by the wrong buffer, or parameters that are not restored changes every later step.  Four
  steps of AggregationStep + nesterov_lookahead on the GPU against the same sequence with the reference's own
  torch-CPU operations (`model.get().sub_(momentum, alpha=momentum * lr)`)."""
  from byzantinemomentum_amd.step import AggregationStep
  from tests.step_reference import ReferenceLoop, assert_floats_close
  n, f, d, mu, lr = 11, 2, 40013, 0.9, 0.05
  h = n - f
  gen = torch.Generator().manual_seed(99)
  a = [0.5 + 0.1 * torch.rand(d, generator=gen) for _ in range(h)]
  b = [0.3 * torch.randn(d, generator=gen) for _ in range(h)]
  step = AggregationStep(n, f, f, gar="krum", momentum=mu, dampening=0.0, momentum_at=momentum_at, attack="empire",
                         attack_factor=1.1, nb_past=2, single_call=False)
  ref = ReferenceLoop(n, f, f, "krum", momentum_at, mu, 0.0, "empire", 1.1, None, 2)
  theta_ref = torch.randn(d, generator=gen)
  origin = theta_ref.clone()
  theta = theta_ref.to(DEV)
  a_dev, b_dev = [t.to(DEV) for t in a], [t.to(DEV) for t in b]
  for it in range(4):
    # ---- reference order of operations, CPU (attack.py:757-783) ----
    sampled_ref = []
    snapshot = theta_ref.clone()                                   # local_chckpt = snapshot(model, deepcopy=True)
    if momentum_at != "worker" and ref.server is not None:
      theta_ref.sub_(ref.server, alpha=(mu * lr))                  # :765
    for i in range(h):
      if momentum_at == "worker" and ref.workers is not None:
        theta_ref.sub_(ref.workers[i], alpha=(mu * lr))            # :770
      sampled_ref.append(a[i] * theta_ref + b[i])                   # grad = model.backprop()
      if momentum_at == "worker":
        theta_ref.copy_(snapshot)                                   # :775 local_chckpt.restore(model)
    if momentum_at != "worker":
      theta_ref.copy_(snapshot)                                     # :783
    # ---- the same with the device path ----
    sampled = []
    keep = theta.clone()
    if momentum_at != "worker":
      step.nesterov_lookahead(theta, lr)
    for i in range(h):
      if momentum_at == "worker":
        step.nesterov_lookahead(theta, lr, worker=i)
      sampled.append(a_dev[i] * theta + b_dev[i])
      if momentum_at == "worker":
        theta.copy_(keep)
    if momentum_at != "worker":
      theta.copy_(keep)
    for g, w in zip(sampled, sampled_ref):
      assert float((g.cpu() - w).abs().max()) <= 2e-6 * float(w.abs().max()), (momentum_at, it)
    want_def, want_upd, want = ref.step(sampled_ref, theta_ref, origin)
    got_def = step.run(sampled, theta, origin.to(DEV))
    scale = float(torch.stack(sampled_ref).abs().max())
    assert float((got_def.cpu() - want_def).abs().max()) <= 4e-6 * scale, (momentum_at, it)
    assert float((step.update_gradient().cpu() - want_upd).abs().max()) <= 4e-6 * scale, (momentum_at, it)
    assert_floats_close(step.floats(), want, tag=(momentum_at, it), tol=1e-5)
    theta_ref = theta_ref - lr * want_upd                           # model.update(...)
    theta = theta - lr * step.update_gradient()
    assert float((theta.cpu() - theta_ref).abs().max()) <= 1e-5 * float(theta_ref.abs().max())
  # the shift itself, against torch's own op on the same GPU
  p = torch.randn(d, device=DEV)
  want = p.sub(step.buffers[1] if momentum_at == "worker" else step.server_momentum, alpha=mu * lr)
  got = step.nesterov_lookahead(p.clone(), lr, worker=1 if momentum_at == "worker" else None)
  assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())


# ---------------------------------------------------------------------------- #
# The Brute subset search on the device (brute.py:47-68)

def test_brute_search_on_the_device_equals_the_host_search(bm):
  """bm_brute_select_device (one wave, squared distances where the distance pass left them) against bm_brute_select
  (host; pinned on exhaustive enumeration by tests/test_host_logic.py): random matrices of many shapes, matrices of
  FEW distinct values (ties everywhere: the lexicographically first subset must come out), zero distances (aliased
  rows), rows at non-finite distance of everything, and the case where no subset has a finite diameter."""
  gars = bm.gars
  gen = torch.Generator().manual_seed(7)
  cases = 0
  for n, f in ((4, 1), (7, 2), (11, 2), (11, 4), (25, 5), (25, 11), (33, 8), (51, 12), (64, 20), (64, 1), (9, 0)):
    for variant in range(8):
      if variant % 4 == 0:
        pts = torch.randn(n, 6, generator=gen, dtype=torch.float64)
      elif variant % 4 == 1:  # few distinct distances
        pts = torch.randint(0, 3, (n, 4), generator=gen).double()
      elif variant % 4 == 2:  # a tight cluster of n - f rows plus outliers, then aliased rows
        pts = torch.randn(n, 5, generator=gen, dtype=torch.float64)
        pts[: n - f] *= 0.01
        pts[-1] = pts[-2]
      else:
        pts = torch.rand(n, 3, generator=gen, dtype=torch.float64).round(decimals=1)
      sq = (pts[:, None, :] - pts[None, :, :]).pow(2).sum(dim=2)
      bad_rows = 0
      if variant >= 4 and f >= 1:  # rows with a non-finite coordinate: at most f of them, then f + 1
        bad_rows = min(f, 2) if variant < 6 else f + 1
        for r in range(bad_rows):
          row = (3 * r + 1) % n
          sq[row, :] = math.nan if r % 2 == 0 else math.inf
          sq[:, row] = sq[row, :]
      sq_dev = sq.to(DEV).contiguous()
      sel, status = gars.brute_select_device(sq_dev, n, f)
      try:
        want = gars.brute_select_host(sq.sqrt().contiguous(), n, f)
      except RuntimeError:
        want = None
      if want is None:
        assert int(status.item()) == -1, (n, f, variant)
        picked = sel[: n - f].tolist()
        assert len(set(picked)) == 1 and not math.isfinite(float(sq[picked[0], (picked[0] + 1) % n])), (n, f, variant)
      else:
        assert int(status.item()) == 0 and sel[: n - f].tolist() == want, (n, f, variant, sel[: n - f].tolist(), want)
        assert sel[n - f:].abs().sum().item() == 0
      cases += 1
  assert cases == 88


def test_brute_rule_runs_without_the_host_and_can_be_graphed(bm):
  """gars.brute: distances -> device search -> selected mean on one stream, the oracle's selection; recorded into a
  HIP graph (impossible with a host search in the middle) the replay gives the same bits, also after the contents
  of the rows changed."""
  from byzantinemomentum_amd.graphs import GraphedCall
  n, f, d = 25, 5, 300007
  rows, h = O.make_stack("hetero", n, f, d, seed=11)
  seen = {}
  dev = [seen.setdefault(id(g), g.to(DEV)) for g in rows]
  want_sel = O.brute_selection(rows, f)
  assert bm.gars.brute_selection(dev, f) == want_sel
  eager = bm.brute(dev, f)
  assert torch.equal(eager.cpu(), O.brute(rows, f))
  graphed = GraphedCall(lambda: bm.brute(dev, f))
  assert torch.equal(graphed(), eager)
  for g in dev[:h]:
    g.mul_(-0.5)
  bm.gars.invalidate_rank_cache()
  assert torch.equal(graphed(), bm.brute(dev, f))


# ---------------------------------------------------------------------------- #
# The evaluate-only form of the factor search (bm_colwise_eval) against the per-evaluation form

@pytest.mark.parametrize("gar,n,f,d,attack,negative", [
  ("trmean", 25, 5, 1000003, "empire", False), ("phocas", 25, 5, 300001, "little", True),
  ("meamed", 25, 11, 262144, "empire", False), ("trmean", 11, 2, 65537, "little", False),
  ("meamed", 11, 4, 4099, "empire", True), ("trmean", 51, 12, 200002, "empire", False),
  ("phocas", 51, 24, 100001, "little", False)])
def test_evaluate_only_search_form_against_the_per_evaluation_form(bm, gar, n, f, d, attack, negative):
  """attacks/identical.py:67-77 against trmean / phocas / meamed: `auto` evaluates every candidate with ONE pass
  that writes nothing (candidate, rule and objective in registers, bm_colwise_eval), `generic` writes the candidate
  vector, runs the rule on the n rows and takes the objective from the distance kernel, like the reference does.
  Same candidates in the same order, objectives within 1e-6 (the two sums round differently), the same factor and
  the same aggregated gradient (the final aggregation is the rule itself in both) — unaligned lengths, a one-,
  two- and three-column tail, n = 11 / 25 / 51, and a column with a NaN."""
  from byzantinemomentum_amd.step import AggregationStep
  h = n - f
  gen = torch.Generator(device=DEV).manual_seed(61)
  base = 0.2 * torch.randn(d, device=DEV, generator=gen)
  honests = [base + (0.5 + 0.05 * i) * torch.randn(d, device=DEV, generator=gen) for i in range(h)]
  assert bm.stats.colwise_eval_supported(gar, n) and not bm.stats.colwise_eval_supported(gar, n + 1)
  traces = {}
  for mode in ("auto", "generic"):
    step = AggregationStep(n, f, f, gar=gar, momentum=0.9, dampening=0.9, momentum_at="update", attack=attack,
                           attack_factor=1.1, nb_past=0, attack_evals=12, attack_negative=negative, line_search=mode)
    out = step.run([g.clone() for g in honests])
    traces[mode] = (step.last_factor, list(step.last_search), out)
  (fa, sa, oa), (fg, sg, og) = traces["auto"], traces["generic"]
  assert len(sa) == len(sg) == 12 and max(y for _, y in sa) > 0
  for (xa, ya), (xg, yg) in zip(sa, sg):
    assert xa == xg and abs(ya - yg) <= 1e-6 * abs(yg) + 1e-30, (gar, xa, ya, yg)
  assert fa == fg and torch.equal(oa, og)
  # the entry point alone, on views that are only 4-byte aligned (the scalar form of the kernel), against the rule
  off = [torch.cat([torch.zeros(1, device=DEV), g])[1:] for g in honests]
  avg = torch.stack(honests).mean(dim=0)
  direction = -avg if attack == "empire" else torch.stack(honests).var(dim=0).sqrt_()
  cand = avg + 0.75 * direction
  want = (getattr(bm, gar)(honests + [torch.addcmul(avg, direction, torch.tensor(0.75, device=DEV))] * f, f).double()
          - avg.double()).pow(2).sum().item()
  got = bm.stats.colwise_eval(gar, off, f, f, avg, direction, 0.75).item()
  assert abs(got - want) <= 1e-5 * want, (got, want, float((cand - avg).norm()))
  # a NaN in one honest row: trmean tolerates up to f of them per column, meamed / phocas follow their centre
  honests[2][5] = math.nan
  got = {}
  for mode in ("auto", "generic"):
    step = AggregationStep(n, f, f, gar=gar, momentum_at="update", attack=attack, nb_past=0, attack_evals=3,
                           attack_negative=negative, line_search=mode)
    step.run([g.clone() for g in honests])
    got[mode] = step.last_search
  for (xa, ya), (xg, yg) in zip(got["auto"], got["generic"]):
    assert xa == xg and ((math.isnan(ya) and math.isnan(yg)) or abs(ya - yg) <= 1e-6 * abs(yg)), (gar, ya, yg)
