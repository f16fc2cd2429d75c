"""The N>1 path on CPU: two processes, gloo backend, each holding half of the coordinates.
Selections must equal the single-process result, coordinate-wise outputs must equal the
corresponding slices, and the all-gathered vector must equal the unsharded aggregation."""

import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gar_oracle as O

N, F, D = 11, 2, 1000


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


def _worker(rank, world, port, queue):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from byzantinemomentum_amd.sharded import ShardedAggregator, shard_bounds
    from tests.sharded_backend import OracleBackend
    rows, h = O.make_stack("hetero", N, F, D, seed=42)
    lo, hi = shard_bounds(D, world, rank)
    seen = {}
    local = []
    for g in rows:  # keep the aliasing of the Byzantine rows
      if id(g) not in seen:
        seen[id(g)] = g[lo:hi].clone()
      local.append(seen[id(g)])
    agg = ShardedAggregator(backend=OracleBackend())
    assert agg.world_size == world
    res = {}
    for name, fn in (("median", lambda: agg.median(local)), ("trmean", lambda: agg.trmean(local, F)),
                     ("krum", lambda: agg.krum(local, F)), ("krum1", lambda: agg.krum(local, F, 1)),
                     ("bulyan", lambda: agg.bulyan(local, F)), ("aksel", lambda: agg.aksel(local, F))):
      res[name] = agg.all_gather_output(fn(), D)
    avg, norm, dev, mx = agg.compute_avg_dev_max(local[:h])
    res["avg"] = agg.all_gather_output(avg, D)
    res["stats"] = (norm, dev, mx)
    res["sq"] = agg.global_sqdist(local)
    # every distance pass of every rank was handed the TRUE total length (not d_local x ranks: the last shard is short)
    res["totals_ok"] = bool(agg.backend.totals_seen) and all(t == D for t in agg.backend.totals_seen)
    # the total is determined once per aggregator; another shard length is refused locally (never a lone collective),
    # an explicit d_total or a reset on every rank serves another vector length
    try:
      agg.total_length((hi - lo) + 64)
      res["refuses_other_length"] = False
    except ValueError:
      res["refuses_other_length"] = True
    res["explicit_total"] = agg.total_length((hi - lo) + 64, d_total=D + 128)
    agg.reset_total_length()
    res["total_after_reset"] = agg.total_length(hi - lo)
    # numpy arrays are pickled by value; torch tensors would be shared through file descriptors of
    # this process, which may already have exited when the parent rebuilds them
    queue.put((rank, {k: (v.detach().numpy().copy() if torch.is_tensor(v) else v) for k, v in res.items()}))
    dist.barrier()
  finally:
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_aggregation_matches_single_process():
  world = 2
  ctx = mp.get_context("spawn")
  queue = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
  for p in procs:
    p.start()
  results = dict(queue.get(timeout=240) for _ in range(world))
  results = {r: {k: (torch.from_numpy(v) if hasattr(v, "dtype") else v) for k, v in res.items()}
             for r, res in results.items()}
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  rows, h = O.make_stack("hetero", N, F, D, seed=42)
  want = {"median": O.median(rows), "trmean": O.trmean(rows, F), "krum": O.krum(rows, F), "krum1": O.krum(rows, F, 1),
          "bulyan": O.bulyan(rows, F), "aksel": O.aksel(rows, F)}
  for r in range(world):
    got = results[r]
    for name, ref in want.items():
      assert got[name].shape == ref.shape
      if name in ("median", "krum", "krum1", "aksel"):
        assert torch.equal(got[name], ref), (r, name)      # same selection, same sequential sums
      else:
        assert torch.allclose(got[name], ref, rtol=0, atol=2e-6), (r, name)
    avg, norm, dev, mx = O.compute_avg_dev_max(rows[:h], "f64")
    assert torch.allclose(got["avg"].double(), avg, atol=1e-6)
    assert abs(got["stats"][0] - norm) <= 1e-6 * norm and abs(got["stats"][1] - dev) <= 1e-6 * dev
    assert abs(got["stats"][2] - mx) <= 1e-6 * mx
    d64 = torch.from_numpy(O.pairwise_distances(rows, "f64")) ** 2
    assert torch.allclose(got["sq"], d64, rtol=1e-12)
  assert torch.equal(results[0]["sq"], results[1]["sq"])   # every rank ranks the same bits
  assert results[0]["totals_ok"] and results[1]["totals_ok"]
  for r in range(world):
    assert results[r]["refuses_other_length"] and results[r]["explicit_total"] == D + 128
    assert results[r]["total_after_reset"] == D


def _a2a_worker(rank, world, port, n, d, queue):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  try:
    from byzantinemomentum_amd.sharded import ShardedAggregator, owned_workers, shard_bounds
    from tests.sharded_backend import OracleBackend
    rows, _ = O.make_stack("iid", n, 0, d, seed=7)          # every rank can regenerate every worker's gradient
    agg = ShardedAggregator(backend=OracleBackend())
    local = agg.to_dim_sharded([rows[i] for i in owned_workers(n, world, rank)], n, d)
    lo, hi = shard_bounds(d, world, rank)
    ok = len(local) == n and all(torch.equal(local[i], rows[i][lo:hi]) for i in range(n))
    med = agg.all_gather_output(agg.median(local), d)       # and the rules run on the exchanged layout
    queue.put((rank, ok, med.numpy().copy()))
    dist.barrier()
  finally:
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n,d", [(5, 1000), (4, 130), (3, 40), (1, 200)])
def test_worker_major_to_dim_major_all_to_all(n, d):
  """Worker-parallel production (rank p holds workers p, p+P, ...) -> one all-to-all -> every rank holds
  its coordinate slice of ALL workers; uneven worker counts, a short last shard, an empty one, and a rank
  that owns no worker at all (n < P: it still joins the exchange, with an all-zero send buffer)."""
  world = 2
  ctx = mp.get_context("spawn")
  queue = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_a2a_worker, args=(r, world, port, n, d, queue)) for r in range(world)]
  for p in procs:
    p.start()
  results = [queue.get(timeout=240) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  rows, _ = O.make_stack("iid", n, 0, d, seed=7)
  for rank, ok, med in results:
    assert ok, f"rank {rank}: exchanged slices differ"
    assert torch.equal(torch.from_numpy(med), O.median(rows))


def test_shard_bounds_cover_everything_once():
  from byzantinemomentum_amd.sharded import shard_bounds
  for d in (1, 63, 64, 65, 1000, 11173962, 36546980):
    for world in (1, 2, 4, 8):
      spans = [shard_bounds(d, world, r) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == d
      for (a, b), (c, e) in zip(spans, spans[1:]):
        assert b == c and a <= b and c <= e
      assert all(lo % 64 == 0 or lo == d for lo, _ in spans)  # empty trailing shards sit at d


def test_single_process_issues_no_collective():
  from byzantinemomentum_amd.sharded import ShardedAggregator
  from tests.sharded_backend import OracleBackend
  assert not dist.is_initialized()
  agg = ShardedAggregator(backend=OracleBackend())
  rows, _ = O.make_stack("hetero", N, F, 257, seed=1)
  assert agg.world_size == 1
  assert torch.equal(agg.krum(rows, F), O.krum(rows, F))
  assert torch.equal(agg.all_gather_output(agg.median(rows), 257), O.median(rows))


def test_one_rank_forced_collectives_serves_every_length():
  """CPU twin of tests/test_gpu_y_rccl_one_rank.py's length loop: ONE aggregator with forced collectives (one-rank
  gloo group) serves vectors of several lengths, because `to_dim_sharded` / `shard_rows` hand out `Shards` that carry
  the total length; plain lists of another length need `d_total=` and are otherwise refused with a message."""
  from byzantinemomentum_amd.sharded import ShardedAggregator, Shards
  from tests.sharded_backend import OracleBackend
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
  dist.init_process_group("gloo", rank=0, world_size=1)
  try:
    rows, h = O.make_stack("hetero", 7, 1, 4100, seed=18)
    agg = ShardedAggregator(backend=OracleBackend(), force_collectives=True)
    assert agg.collective
    assert torch.equal(agg.krum(rows, 1), O.krum(rows, 1))          # a plain list: the total is determined once (4100)
    for d_odd in (4007, 64, 1):
      grads = [g[:d_odd].contiguous() for g in rows]
      local = agg.to_dim_sharded(grads, 7, d_odd)
      assert isinstance(local, Shards) and local.d_total == d_odd and len(local) == 7
      assert all(torch.equal(a, b) for a, b in zip(local, grads))
      assert torch.equal(agg.median(local), O.median(grads))
      assert torch.equal(agg.krum(local, 1), O.krum(grads, 1))
      assert torch.allclose(agg.bulyan(local, 1), O.bulyan(grads, 1), rtol=0, atol=2e-6)
      assert torch.equal(agg.brute(local, 1), O.brute(grads, 1))
      assert agg.backend.totals_seen[-1] == d_odd
    short = [g[:100].contiguous() for g in rows]
    with pytest.raises(ValueError, match="d_total"):
      agg.krum(short, 1)
    assert torch.equal(agg.krum(short, 1, d_total=100), O.krum(short, 1))
    assert torch.equal(agg.krum(agg.shard_rows(short), 1), O.krum(short, 1))
    assert torch.equal(agg.krum(rows, 1), O.krum(rows, 1))          # the first length is still served
  finally:
    dist.destroy_process_group()
