"""Host logic of byzantinemomentum_amd.step.AggregationStep on CPU: every momentum placement, clipping,
both attacks and several rules against the independent loop of oracle/step_oracle.py, with the
oracle-backed compute legs; then the same step dim-sharded over two gloo ranks against one rank."""

import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gar_oracle as O
from tests.step_reference import ReferenceLoop, assert_floats_close

N, F, D = 11, 2, 1536


def sampled_for_step(it, h, d=D, extra=0):
  gen = torch.Generator().manual_seed(1000 + it)
  base = 0.2 * torch.randn(d, generator=gen)
  return [base + (0.5 + 0.1 * i) * torch.randn(d, generator=gen) for i in range(h + extra)]


CONFIGS = [
  dict(gar="krum", momentum_at="worker", clip=None, attack="empire", factor=1.1),
  dict(gar="krum", momentum_at="worker", clip=30.0, attack="empire", factor=1.1),
  dict(gar="bulyan", momentum_at="server", clip=None, attack="little", factor=1.5),
  dict(gar="median", momentum_at="update", clip=32.0, attack="empire", factor=1.1),
  dict(gar="trmean", momentum_at="server", clip=28.0, attack="little", factor=-1.5),
  dict(gar="aksel", momentum_at="update", clip=None, attack="empire", factor=1.1),
  dict(gar="brute", momentum_at="worker", clip=None, attack="empire", factor=1.1),
  dict(gar="average", momentum_at="worker", clip=None, attack="empire", factor=1.1),
  dict(gar="cge", momentum_at="worker", clip=None, attack="empire", factor=1.1),
]


# the factor search of the attacks (identical.py:67-77): evals = E is the reference's `factor:-E`
SEARCH_CONFIGS = [
  dict(gar="krum", momentum_at="worker", clip=None, attack="empire", factor=1.1, evals=16),
  dict(gar="krum", momentum_at="worker", clip=None, attack="little", factor=1.1, evals=16, line_search="generic"),
  dict(gar="krum", momentum_at="server", clip=30.0, attack="little", factor=1.1, evals=9, negative=True),
  dict(gar="brute", momentum_at="update", clip=None, attack="little", factor=1.1, evals=8),
  dict(gar="average", momentum_at="worker", clip=None, attack="empire", factor=1.1, evals=5, negative=True),
  dict(gar="median", momentum_at="update", clip=None, attack="empire", factor=1.1, evals=16),
  dict(gar="bulyan", momentum_at="server", clip=None, attack="little", factor=1.1, evals=6),
  dict(gar="trmean", momentum_at="worker", clip=None, attack="empire", factor=1.1, evals=1),
]


def config_id(c):
  tag = f"{c['gar']}-{c['momentum_at']}-clip{c['clip']}-{c['attack']}"
  if "evals" in c:
    tag += f"-search{c['evals']}{'neg' if c.get('negative') else ''}-{c.get('line_search', 'auto')}"
  return tag


def reference_for(cfg, n=N, f=F):
  return ReferenceLoop(n, f, f, cfg["gar"], cfg["momentum_at"], 0.9, 0.9, cfg["attack"], cfg["factor"], cfg["clip"], 3,
                       evals=cfg.get("evals"), negative=cfg.get("negative", False))


def assert_same_search(step, ref, tag):
  """Same factor, or a tie: the objective values the two searches hold at their factors agree within 1e-5
  (a comparison between two candidates that close is decided by rounding in the reference as well)."""
  got, want = step.last_search, ref.last_search
  assert len(got) == len(want), tag
  floor = 1e-9 * max(y for _, y in want)  # an objective that is zero up to rounding (the rule returned the honest mean)
  if step.last_factor == ref.last_factor:
    for (x, y), (xo, yo) in zip(got, want):
      assert x == xo and abs(y - yo) <= 1e-5 * abs(yo) + floor, (tag, x, y, yo)
    return
  best_got = max(y for _, y in got)
  best_want = max(y for _, y in want)
  assert abs(best_got - best_want) <= 1e-5 * max(abs(best_want), 1e-9), (tag, step.last_factor, ref.last_factor)


def make_step(cfg, aggregator=None, n=N, f=F):
  from byzantinemomentum_amd.sharded import ShardedAggregator
  from byzantinemomentum_amd.step import AggregationStep
  from tests.sharded_backend import OracleBackend
  agg = aggregator or ShardedAggregator(backend=OracleBackend())
  return AggregationStep(n, f, f, gar=cfg["gar"], momentum=0.9, dampening=0.9, momentum_at=cfg["momentum_at"],
                         attack=cfg["attack"], attack_factor=cfg["factor"], nb_past=3, gradient_clip=cfg["clip"],
                         aggregator=agg, attack_evals=cfg.get("evals"), attack_negative=cfg.get("negative", False),
                         line_search=cfg.get("line_search", "auto"))


@pytest.mark.parametrize("cfg", CONFIGS + SEARCH_CONFIGS, ids=config_id)
def test_step_matches_reference_loop(cfg):
  assert not dist.is_initialized()
  h = N - F
  n = N
  step = make_step(cfg)
  ref = reference_for(cfg)
  gen = torch.Generator().manual_seed(5)
  origin = torch.randn(D, generator=gen)
  params = origin.clone()
  for it in range(4):
    sampled = sampled_for_step(it, h, extra=(1 if it == 2 else 0))  # one step with a gradient sampled only for the study
    want_def, want_upd, want = ref.step(sampled, params, origin)
    got_def = step.run([g.clone() for g in sampled], params, origin)
    if "evals" in cfg:
      assert_same_search(step, ref, (config_id(cfg), it))
      if step.last_factor != ref.last_factor:
        pytest.skip("the two searches settled on tied candidates")  # (never hit with these seeds)
    if it == 1:
      step.run  # noqa: B018  (floats() is skipped on this step: the past deque must still advance)
    else:
      got = step.floats()
      assert step.floats() is got                      # idempotent
      assert_floats_close(got, want, tag=(cfg["gar"], it), tol=2e-6)
    scale = float(torch.stack(sampled).abs().max())
    assert float((got_def - want_def).abs().max()) <= 2e-6 * scale, (cfg, it)
    assert float((step.update_gradient() - want_upd).abs().max()) <= 2e-6 * scale, (cfg, it)
    params = params - 0.05 * want_upd


def test_step_rejects_bad_arguments():
  from byzantinemomentum_amd.step import AggregationStep, MAX_PAST
  from byzantinemomentum_amd.sharded import ShardedAggregator
  from tests.sharded_backend import OracleBackend
  agg = ShardedAggregator(backend=OracleBackend())
  with pytest.raises(ValueError):
    AggregationStep(11, 2, 2, gar="nope", aggregator=agg)
  with pytest.raises(ValueError):
    AggregationStep(11, 2, 2, momentum_at="client", aggregator=agg)
  with pytest.raises(ValueError):
    AggregationStep(11, 2, 2, nb_past=MAX_PAST + 1, aggregator=agg)
  with pytest.raises(ValueError):
    AggregationStep(11, 2, 2, attack_evals=0, aggregator=agg)
  with pytest.raises(ValueError):
    AggregationStep(11, 2, 2, attack_evals=4, line_search="fast", aggregator=agg)
  step = AggregationStep(11, 2, 2, aggregator=agg)
  with pytest.raises(ValueError):
    step.run([torch.zeros(8)] * 3)
  with pytest.raises(RuntimeError):
    step.floats()


# ---------------------------------------------------------------------------- #
# Two gloo ranks, each holding a slice of the coordinates (one of them may be short or empty)

def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


SHARDED_CONFIGS = CONFIGS[:6] + [SEARCH_CONFIGS[0], SEARCH_CONFIGS[5]]  # + the search, scalar and per-evaluation forms


def _worker(rank, world, port, d, queue):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from byzantinemomentum_amd.sharded import ShardedAggregator, shard_bounds
    from tests.sharded_backend import OracleBackend
    lo, hi = shard_bounds(d, world, rank)
    out = {}
    for ci, cfg in enumerate(SHARDED_CONFIGS):
      agg = ShardedAggregator(backend=OracleBackend())
      assert agg.world_size == world and agg.collective
      step = make_step(cfg, agg)
      gen = torch.Generator().manual_seed(5)
      origin = torch.randn(d, generator=gen)
      params = origin + 0.01
      for it in range(3):
        sampled = [g[lo:hi].clone() for g in sampled_for_step(it, N - F, d)]
        defense = step.run(sampled, params[lo:hi].clone(), origin[lo:hi].clone())
        floats = step.floats()
        out[(ci, it)] = (agg.all_gather_output(defense, d).numpy().copy(), floats)
    queue.put((rank, out))
    dist.barrier()
  finally:
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,d", [(2, D), (2, 100), (2, 40), (3, D), (3, 130), (4, 300)])
def test_sharded_step_matches_single_rank(world, d):
  """d = 100 / world 2: the second rank's shard is short (36 coordinates); d = 40: it is EMPTY;
  world 3, d = 130: shards of 64, 64 and 2 coordinates; world 4, d = 300: 128, 128, 44 and an empty one."""
  ctx = mp.get_context("spawn")
  queue = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, d, queue)) for r in range(world)]
  for p in procs:
    p.start()
  results = dict(queue.get(timeout=500) for _ in range(world))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for ci, cfg in enumerate(SHARDED_CONFIGS):
    single = make_step(cfg)
    gen = torch.Generator().manual_seed(5)
    origin = torch.randn(d, generator=gen)
    params = origin + 0.01
    for it in range(3):
      want_def = single.run([g.clone() for g in sampled_for_step(it, N - F, d)], params, origin)
      want = single.floats()
      for r in range(world):
        got_def, got = results[r][(ci, it)]
        assert torch.equal(torch.from_numpy(got_def), want_def) or \
            float((torch.from_numpy(got_def) - want_def).abs().max()) <= 1e-6, (cfg, it, r)
        for key, val in want.items():
          g = got[key]
          assert (math.isnan(g) and math.isnan(val)) or abs(g - val) <= 1e-9 * max(abs(val), 1e-6), (cfg, it, key, g, val)
      # every rank computed the same floats (same packed exchange, same reduction order)
      a = results[0][(ci, it)][1]
      for r in range(1, world):
        b = results[r][(ci, it)][1]
        assert all(a[k] == b[k] or (math.isnan(a[k]) and math.isnan(b[k])) for k in a)


def test_empty_shard_layout():
  """d < 64: the second rank's slice is EMPTY (the d = 40 case above runs the whole step on it: no rank
  may skip a collective)."""
  from byzantinemomentum_amd.sharded import shard_bounds
  assert shard_bounds(40, 2, 1) == (40, 40)


@pytest.mark.parametrize("n,f,attack,negative", [(11, 2, "empire", False), (11, 5, "little", True), (9, 5, "empire", False),
                                                 (25, 5, "little", False)])
def test_median_factor_search_from_two_order_statistics(n, f, attack, negative):
  """The factor search against the median (attacks/identical.py:67-77): `auto` evaluates every candidate as the middle
  of (candidate, lo, hi) — lo / hi = medians of the honest rows with f copies of -inf / +inf — instead of the median
  of all n rows.  Same candidates, same objective values BIT FOR BIT as the per-evaluation form, on columns with
  ties, infinities and NaN too, and with a Byzantine majority (lo = -inf, hi = +inf everywhere)."""
  from byzantinemomentum_amd.sharded import ShardedAggregator
  from byzantinemomentum_amd.step import AggregationStep
  from tests.sharded_backend import OracleBackend
  h = n - f
  runs = {}
  for mode in ("auto", "generic"):
    step = AggregationStep(n, f, f, gar="median", momentum=0.9, dampening=0.9, momentum_at="worker", attack=attack,
                           attack_factor=1.1, nb_past=2, aggregator=ShardedAggregator(backend=OracleBackend()),
                           attack_evals=12, attack_negative=negative, line_search=mode)
    trace = []
    for it in range(3):
      sampled = sampled_for_step(it, h, d=701)
      for g in sampled:
        g[::7] = g[::7].round()        # exact ties between workers
      if it == 2:  # (a non-finite honest value makes the attack's average, hence the objective, non-finite: last step only)
        sampled[1][5] = math.inf
        sampled[2][6] = -math.inf
        sampled[0][11] = math.nan
      out = step.run(sampled)
      trace.append((step.last_factor, list(step.last_search), out.clone()))
    runs[mode] = trace
  for (fa, sa, oa), (fg, sg, og) in zip(runs["auto"], runs["generic"]):
    assert len(sa) == len(sg) == 12
    for (xa, ya), (xg, yg) in zip(sa, sg):
      assert xa == xg and (ya == yg or (math.isnan(ya) and math.isnan(yg))), (xa, ya, yg)
    assert fa == fg
    assert torch.equal(oa.nan_to_num(nan=7.0), og.nan_to_num(nan=7.0))
  first = [y for _, y in runs["auto"][0][1]]  # the first steps compare real objective values
  assert all(math.isfinite(y) and y >= 0 for y in first) and max(first) > 0


@pytest.mark.parametrize("n,f,attack,negative,m", [(11, 2, "empire", False, None), (15, 3, "little", True, None),
                                                   (25, 5, "little", False, None), (11, 2, "empire", False, 3)])
def test_bulyan_factor_search_ranks_from_scalars(n, f, attack, negative, m):
  """The factor search against Bulyan: `auto` ranks every candidate stack on the host from the inner products of ONE
  distance pass (bm_attack_ranking) and runs only the rule's second pass on the vectors; `generic` runs the whole
  rule per evaluation.  Equal rankings give the same vectors, hence the same candidates and objective values."""
  from byzantinemomentum_amd.sharded import ShardedAggregator
  from byzantinemomentum_amd.step import AggregationStep
  from tests.sharded_backend import OracleBackend
  h = n - f
  runs = {}
  for mode in ("auto", "generic"):
    step = AggregationStep(n, f, f, gar="bulyan", gar_args={} if m is None else {"m": m}, momentum=0.9, dampening=0.9,
                           momentum_at="server", attack=attack, attack_factor=1.1, nb_past=2,
                           aggregator=ShardedAggregator(backend=OracleBackend()), attack_evals=10,
                           attack_negative=negative, line_search=mode)
    trace = []
    for it in range(3):
      out = step.run(sampled_for_step(it, h, d=801))
      trace.append((step.last_factor, list(step.last_search), out.clone()))
    runs[mode] = trace
  for (fa, sa, oa), (fg, sg, og) in zip(runs["auto"], runs["generic"]):
    assert fa == fg and sa == sg and len(sa) == 10
    assert torch.equal(oa, og)
  assert max(y for _, y in runs["auto"][0][1]) > 0
