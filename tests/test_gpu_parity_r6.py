"""Round 6: the failure paths the round-5 review found open.

* The device Brute search that runs out of its node budget (status -2) must never leave a usable selection behind
  (aggregators/brute.py:47-80 would keep computing): unchecked, the aggregate is NaN everywhere; checked — what the
  `native-brute` plugin and AggregationStep do — the host search, which has no budget, answers instead.
* `BM_PAIR_TAU <= 0` (the accuracy gate off) must still rank: nothing may be listed for an exact pass that is not launched.
* n = 64 with every row listed by the gate: the gated direct kernel ranks inside its own launch with 64 KB of dynamic LDS
  on top of its static LDS (one opt-in rule for every kernel, bm_common.h lds_opt_in).

Each case runs in a process of its own: the knobs are process-global (bm_tuning_set is test-only, include/bm_gar.h)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(body, env=None):
  full = dict(os.environ)
  full.update(env or {})
  head = "import sys, math, torch\nsys.path.insert(0, %r)\ndev = 'cuda:0'\n" % ROOT
  done = subprocess.run([sys.executable, "-c", head + body], cwd=ROOT, capture_output=True, text=True, timeout=600, env=full)
  assert done.returncode == 0, (done.stdout[-2000:], done.stderr[-4000:])
  return done.stdout


def test_brute_budget_exhaustion_never_yields_a_usable_selection():
  _run('''
import byzantinemomentum_amd as bm
from byzantinemomentum_amd import _lib, gars
from byzantinemomentum_amd.sharded import ShardedAggregator
from byzantinemomentum_amd.step import AggregationStep
from oracle import gar_oracle as O
lib = _lib.load()
n, f, d = 11, 2, 5003
rows, h = O.make_stack("hetero", n, f, d, seed=3)
seen = {}
dv = [seen.setdefault(id(g), g.to(dev)) for g in rows]
want = bm.brute(dv, f)
assert int(want.brute_status.item()) == 0 and torch.equal(want.cpu(), O.brute(rows, f))
gars.invalidate_rank_cache()
assert lib.bm_tuning_set(b"BM_BRUTE_BUDGET", 1) == 0          # one search-tree node per wave: every search gives up
try:
    out = bm.brute(dv, f)                                       # unchecked: asynchronous, and visibly unusable
    assert int(out.brute_status.item()) == -2
    assert bool(out.isnan().all()), "status -2 must not leave an average of some rows"
    try:
        gars.brute_check(out.brute_status)
        raise SystemExit("brute_check must raise on -2")
    except RuntimeError as err:
        assert "budget" in str(err)
    gars.invalidate_rank_cache()
    checked = bm.brute(dv, f, check=True)                       # the plugin's path: the host search answers
    assert int(checked.brute_status.item()) == 0 and torch.equal(checked, want)
    gars.invalidate_rank_cache()
    assert gars.brute_selection(dv, f) == list(O.brute_selection(rows, f))   # (host search behind a device search that gave up)
    gars.invalidate_rank_cache()
    # the sharded aggregator keeps its own status; the step checks before it hands out the defense vector
    agg = ShardedAggregator(local_only=True)
    assert bool(agg.brute(dv, f).isnan().all()) and int(agg.brute_status.item()) == -2
    assert torch.equal(agg.brute(dv, f, check=True), want) and agg.brute_status is None
    step = AggregationStep(n, f, f, gar="brute", momentum=0.9, dampening=0.9, attack_factor=1.1, nb_past=2, aggregator=agg)
    defense = step.run(dv[:h])
    floats = step.floats()
    assert bool(defense.isfinite().all()) and math.isfinite(floats["defense_norm_avg"])
finally:
    lib.bm_tuning_set(b"BM_BRUTE_BUDGET", 0)
gars.invalidate_rank_cache()
same = AggregationStep(n, f, f, gar="brute", momentum=0.9, dampening=0.9, attack_factor=1.1, nb_past=2,
                       aggregator=ShardedAggregator(local_only=True)).run(dv[:h])
assert torch.equal(same, defense), "the host fallback and the device search must select the same rows"
print("ok")
''')


def test_ranking_with_the_accuracy_gate_off():
  """BM_PAIR_TAU=0: rounding between near-identical rows leaves some Gram distances slightly negative; they must not be
  listed (no exact pass follows), and the ranking must come out of the reduction's own launch."""
  _run('''
import byzantinemomentum_amd as bm
from byzantinemomentum_amd import gars
from oracle import gar_oracle as O
for n, f, d in ((25, 5, 100003), (51, 12, 40001)):
    gen = torch.Generator().manual_seed(n)
    base = torch.randn(d, generator=gen)
    rows = [base + 1e-4 * torch.randn(d, generator=gen) for _ in range(n - f)]     # near-identical honest rows
    rows += [base * 1.0] * f                                                        # aliased copies
    seen = {}
    dv = [seen.setdefault(id(g), g.to(dev)) for g in rows]
    poison = torch.full((64,), -7, dtype=torch.int32, device=dev)                    # what an unwritten ranking would show
    del poison
    order = gars.krum_selection(dv, f)
    assert len(set(order)) == n - f - 2 and all(0 <= i < n for i in order), order
    out = bm.krum(dv, f)
    assert bool(out.isfinite().all())
    ranking = gars.bulyan_ranking(dv, f) if n >= 4 * f + 3 else None
    assert ranking is None or sorted(ranking) == list(range(n)), ranking
print("ok")
''', env={"BM_PAIR_TAU": "0"})


def test_n64_with_every_row_listed_ranks_inside_the_gated_kernel():
  """BM_PAIR_TAU=1e30 lists every row: at n = 64 the gated direct kernel recomputes all pairs and ranks with 64 KB of
  dynamic LDS next to its 4 KB of static LDS.  Selections against the oracle's on the same stack."""
  _run('''
import byzantinemomentum_amd as bm
from byzantinemomentum_amd import gars
from oracle import gar_oracle as O
for n, f in ((64, 15), (56, 13)):
    rows, h = O.make_stack("hetero", n, f, 30011, seed=n)
    seen = {}
    dv = [seen.setdefault(id(g), g.to(dev)) for g in rows]
    assert gars.krum_selection(dv, f) == O.krum_order(rows, f)[0][:n - f - 2], n
    assert torch.equal(bm.krum(dv, f).cpu(), O.krum(rows, f)), n
    assert gars.bulyan_ranking(dv, f) == O.bulyan_order(rows, f)[0], n
print("ok")
''', env={"BM_PAIR_TAU": "1e30"})


@pytest.mark.parametrize("burst", ["0", "1"])
def test_study_block_carries_the_momentum_of_the_update(burst):
  """bm_study_stats_update against bm_study_stats followed by bm_multi_fma3(M, M, defense, mu, 1 - damp) (what round 5
  launched, attack.py:836-838): the SAME bits in M, the same statistics (1e-6 relative: the burst form with a second
  staged stream folds its fp32 partial sums into fp64 at other points), the same C — plain form and burst form (BM_STUDY_BURST=1
  forces it), every curvature mode, with and without attack / l2, ragged lengths and an unaligned momentum."""
  _run('''
import byzantinemomentum_amd as bm
from byzantinemomentum_amd import stats
gen = torch.Generator().manual_seed(11)
for d in (5, 4096, 40007, 1 << 20, (1 << 20) + 3):
    for f_real in (0, 3):
        for mode in (0, 1, 2, 3):
            for l2 in (False, True):
                mk = lambda: torch.randn(d, generator=gen).to(dev)
                s, h, df, byz, past, old, par, org = (mk() for _ in range(8))
                curv_a = mk(); curv_b = curv_a.clone()
                slab = mk() if d != 40007 else torch.randn(d + 1, generator=gen).to(dev)[1:]   # (4-byte aligned only)
                mom_a = slab.clone() if d != 40007 else slab
                mom_b = mom_a.clone()
                kw = dict(past_newest=past if mode >= 2 else None, past_oldest=old if mode == 3 else None, curv_mode=mode,
                          mu=0.9, oldest_weight=-(0.9 ** 3), params=par if l2 else None, origin=org if l2 else None)
                want = stats.study_stats(s, h, df, byz if f_real else None, f_real, curv=curv_a if mode >= 1 else None, **kw)
                stats.multi_fma3([mom_a], [mom_a], [df], 0.9, 0.01)
                got = stats.study_stats(s, h, df, byz if f_real else None, f_real, curv=curv_b if mode >= 1 else None,
                                        update_momentum=mom_b, update_mu=0.9, update_omd=0.01, **kw)
                assert torch.equal(mom_a, mom_b), (d, f_real, mode, l2, "momentum bits")
                assert torch.equal(curv_a, curv_b), (d, f_real, mode, l2, "C")
                w, g = want.cpu(), got.cpu()
                scale = w.abs().clamp_min(1e-30)
                assert bool(((w - g).abs() <= 1e-6 * w.abs() + 1e-9 * scale.max()).all()), (d, f_real, mode, l2, (w - g).abs().max().item())
print("ok")
''', env={"BM_STUDY_BURST": burst})
