"""Two to four RCCL ranks, one per GPU: the dim-sharded path with REAL peers — libbm_gar's own communicator
(bm_comm_init with nranks > 1, the fp64 all-reduce inside the single-call rules, bm_allgather_f32), the
torch.distributed form of the same rules, the packed statistics exchange, all_to_all_single with real splits, and a
full step.  SURVEY.md section 8e; BASELINE.json configs[3], configs[4].

`gpurun` and the driver's test box hand out ONE GPU, so the cases with 2 and 4 ranks are SKIPPED there; they are the
vehicle for a node with several GPUs.  The rank body itself does run on every box: the first case is the SAME function
with world = 1 — a one-rank RCCL group with forced collectives (every all-reduce, all-gather and all-to-all goes through
RCCL, alone) — so that on the day a multi-GPU node appears only `nranks` changes.  It sorts late among the GPU tests on
purpose.
"""

import math
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gar_oracle as O

pytestmark = pytest.mark.gpu

N, F = 25, 5


def _close(a, b, tol, what):
  scale = max(float(b.abs().max()) if b.numel() else 0.0, 1e-30)
  err = float((a - b).abs().max()) if b.numel() else 0.0
  assert err <= tol * scale, (what, err, scale)


def _worker(rank, world, rendezvous, d, queue):
  import faulthandler
  import sys
  import traceback
  log = open(os.path.join(os.path.dirname(rendezvous), f"rank{rank}.stderr"), "w", buffering=1)
  os.dup2(log.fileno(), 2)
  sys.stderr = log
  faulthandler.enable(file=log, all_threads=True)
  try:
    _rank_body(rank, world, rendezvous, d, queue)
  except BaseException:  # noqa: BLE001
    traceback.print_exc(file=log)
    queue.put((rank, {"error": traceback.format_exc()}))
    raise


def _rank_body(rank, world, rendezvous, d, queue):
  import datetime
  dev = torch.device("cuda", rank)
  torch.cuda.set_device(dev)
  os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  dist.init_process_group("nccl", init_method=f"file://{rendezvous}", rank=rank, world_size=world,
                          timeout=datetime.timedelta(seconds=300), device_id=dev)
  try:
    import byzantinemomentum_amd as bm
    from byzantinemomentum_amd.sharded import ShardedAggregator, owned_workers, shard_bounds
    from byzantinemomentum_amd.step import AggregationStep

    def shard(rows, lo, hi):
      seen = {}
      return [seen.setdefault(id(g), g[lo:hi].to(dev)) for g in rows]

    lo, hi = shard_bounds(d, world, rank)
    report = {"shard": (lo, hi)}
    forced = dict(force_collectives=True) if world == 1 else {}   # (one rank: the collectives are issued all the same)
    native = ShardedAggregator(**forced)                      # libbm_gar's own RCCL communicator: one C call per rule
    plain = ShardedAggregator(native_comm=False, **forced)    # the same rules through torch.distributed
    assert native.world_size == world and native.collective and plain.native is None and not plain.single_call
    report["native_comm"] = native.native is not None
    rows, h = O.make_stack("hetero", N, F, d, seed=1234)
    local, full = shard(rows, lo, hi), shard(rows, 0, d)
    for tag, agg in (("native", native), ("plain", plain)):
      assert agg.total_length(hi - lo) == d
      sq = agg.global_sqdist(local)
      want_sq = bm.gars.pairwise_sqdist(full)
      off = ~torch.eye(N, dtype=torch.bool, device=dev) & (want_sq > 0)
      assert float(((sq - want_sq).abs()[off] / want_sq[off]).max()) <= 1e-6, tag
      assert float(sq[h, h + 1]) == 0.0
      for name, sharded, single in (("krum", lambda: agg.krum(local, F), lambda: bm.krum(full, F)),
                                    ("krum m=1", lambda: agg.krum(local, F, 1), lambda: bm.krum(full, F, 1)),
                                    ("bulyan", lambda: agg.bulyan(local, F), lambda: bm.bulyan(full, F)),
                                    ("aksel", lambda: agg.aksel(local, F), lambda: bm.aksel(full, F)),
                                    ("cge", lambda: agg.cge(local, F), lambda: bm.cge(full, F)),
                                    ("brute", lambda: agg.brute(local, F), lambda: bm.brute(full, F))):
        got, want = sharded(), single()
        assert got.shape[0] == hi - lo
        if name == "bulyan":  # pass 2 may keep either of two exactly tied deviations
          bad = (got - want[lo:hi]).abs() > 2e-6 * float(want.abs().max())
          assert int(bad.sum()) <= max(1, (hi - lo) // 10000), (tag, name, int(bad.sum()))
        else:
          assert torch.equal(got, want[lo:hi]), (tag, name)
        whole = agg.all_gather_output(got, d)
        assert whole.shape[0] == d and torch.equal(whole[lo:hi], got), (tag, name)
        if name == "krum":  # every rank holds the same whole vector
          report[(tag, "krum checksum")] = float(whole.double().sum())
      assert torch.equal(agg.median(local), bm.median(full)[lo:hi])
      avg, norm, devi, mx = agg.compute_avg_dev_max(local[:h])
      wavg, wnorm, wdev, wmx = bm.compute_avg_dev_max(full[:h])
      assert torch.equal(avg, wavg[lo:hi])
      assert abs(norm - wnorm) <= 1e-9 * wnorm and abs(devi - wdev) <= 1e-9 * wdev and mx == wmx
      report[(tag, "stats")] = (norm, devi, mx)
    # worker-parallel production -> dimension-major through a REAL all-to-all
    prod, _ = O.make_stack("iid", N, 0, d, seed=77)
    mine = [prod[i].to(dev) for i in owned_workers(N, world, rank)]
    got = native.to_dim_sharded(mine, N, d, device=dev)
    assert len(got) == N
    for i in range(N):
      assert torch.equal(got[i].cpu(), prod[i][lo:hi]), i
    if hi > lo:
      assert torch.equal(native.median(got), bm.median(shard(prod, 0, d))[lo:hi])
    # the full step on the slice against the single-rank step on the whole vectors
    for gar in ("krum", "median"):
      sharded = AggregationStep(N, F, F, gar=gar, momentum=0.9, dampening=0.9, attack_factor=1.1, nb_past=2, aggregator=native)
      single = AggregationStep(N, F, F, gar=gar, momentum=0.9, dampening=0.9, attack_factor=1.1, nb_past=2,
                               aggregator=ShardedAggregator(local_only=True))
      gen = torch.Generator().manual_seed(5)
      origin = torch.randn(d, generator=gen)
      params = origin + 0.01
      for it in range(3):
        base = 0.2 * torch.randn(d, generator=gen)
        sampled = [base + (0.5 + 0.05 * i) * torch.randn(d, generator=gen) for i in range(N - F)]
        got_def = sharded.run([g[lo:hi].to(dev) for g in sampled], params[lo:hi].to(dev), origin[lo:hi].to(dev))
        got_f = sharded.floats()
        want_def = single.run([g.to(dev) for g in sampled], params.to(dev), origin.to(dev))
        want_f = single.floats()
        if gar == "median":
          assert torch.equal(got_def, want_def[lo:hi]), (gar, it)
        else:
          _close(got_def, want_def[lo:hi], 2e-6, (gar, it))
        for key, val in want_f.items():
          g = got_f[key]
          assert (math.isnan(g) and math.isnan(val)) or abs(g - val) <= 1e-6 * max(abs(val), 1e-6), (gar, it, key, g, val)
        report[(gar, it)] = tuple(sorted((k, v) for k, v in got_f.items() if not math.isnan(v)))
    torch.cuda.synchronize()
    queue.put((rank, report))
    dist.barrier()
  finally:
    dist.destroy_process_group()


def _visible_gpus():
  return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,d", [(1, 40007), (2, 200003), (4, 1 << 20)])
def test_rccl_ranks_on_separate_gpus(world, d):
  if _visible_gpus() < world:
    pytest.skip(f"{world} GPUs needed, {_visible_gpus()} visible (gpurun boxes have one)")
  import queue as queue_mod
  import tempfile
  import time
  ctx = mp.get_context("spawn")
  queue = ctx.Queue()
  rendezvous = os.path.join(tempfile.mkdtemp(prefix="bm_rccl_ranks_"), "store")
  procs = [ctx.Process(target=_worker, args=(r, world, rendezvous, d, queue)) for r in range(world)]
  for p in procs:
    p.start()
  results, deadline = {}, time.time() + 800
  try:
    while len(results) < world and time.time() < deadline:
      try:
        rank, rep = queue.get(timeout=5)
        results[rank] = rep
      except queue_mod.Empty:
        if any(p.exitcode not in (None, 0) for p in procs):
          break
  finally:
    for p in procs:
      p.join(timeout=30 if len(results) == world else 1)
      if p.is_alive():
        p.terminate()
        p.join(timeout=10)
  words = []
  for r in range(world):
    try:
      with open(os.path.join(os.path.dirname(rendezvous), f"rank{r}.stderr")) as fh:
        text = fh.read().strip()
      if text:
        words.append(f"--- stderr of rank {r} ---\n{text[-4000:]}")
    except OSError:
      pass
  errors = {r: rep["error"] for r, rep in results.items() if "error" in rep}
  codes = [p.exitcode for p in procs]
  assert not errors and len(results) == world and all(c == 0 for c in codes), (
    f"exit codes {codes}\n" + "\n".join(f"--- rank {r} ---\n{t}" for r, t in sorted(errors.items())) + "\n" + "\n".join(words))
  # every rank decoded the same floats, and gathered the same whole vector
  keys = [k for k in results[0] if k != "shard"]
  for r in range(1, world):
    for k in keys:
      assert results[r][k] == results[0][k], (r, k)
  assert results[0]["native_comm"], "libbm_gar's own RCCL communicator was not created (fell back to torch.distributed)"
