"""Several ranks of the dim-sharded path with the HIP kernels underneath — on ONE MI355X.

(Collected last among the GPU tests: four or five processes share the one GPU here — the condition under which, until
round 6, Bulyan's second pass returned a few wrong coordinates in ~1.5 % of its launches: packed fp32 instructions,
DESIGN 8.  The library is built without them; a mismatch here is a failure, never an expected one, and nothing is
retried.)

`gpurun` hands out one GPU, so no RCCL job of more than one rank can run there; the gloo tests cover the
partitioning logic with an oracle backend on CPU.  What neither covers is the REAL kernels on ragged and empty
shards inside a multi-rank job.  Here 2-4 processes share `cuda:0`, each holds its coordinate slice as GPU tensors
and runs `ShardedAggregator(native_comm=False)` with the product's HipBackend; only the transport of the (tiny)
exchanges is replaced: the three collective primitives of the class are staged through host tensors over gloo.
Every rank compares its slice with the UNSHARDED HIP call on the full vectors (which it can make itself: the
GPU is shared), so the kernels are checked against themselves across world sizes and, through the other GPU
tests, against the oracle.  SURVEY.md §8e; BASELINE.json configs[3], configs[4].
"""

import math
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gar_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
N, F = 25, 5


def _staged_aggregator():
  from byzantinemomentum_amd.sharded import ShardedAggregator

  class HostStaged(ShardedAggregator):
    """The product's sharded rules; the bytes of each exchange travel through host memory (gloo)."""

    def _all_reduce(self, tensor, op=None):
      if self.collective:
        host = tensor.cpu()
        dist.all_reduce(host, op=(op or dist.ReduceOp.SUM), group=self.group)
        tensor.copy_(host)
      return tensor

    def _all_gather_into(self, everyone, mine):
      parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(self.world_size)]
      dist.all_gather(parts, mine.cpu(), group=self.group)
      everyone.copy_(torch.cat([p.reshape(-1) for p in parts]).to(everyone.device))

    def _all_to_all(self, recv, send):
      world = self.world_size
      src = list(send.cpu().chunk(world))
      dst = [torch.empty_like(s) for s in src]
      # gloo has no all_to_all: P broadcasts of what each rank sends, every receiver keeps its own part
      for r in range(world):
        box = [s.clone() for s in src] if r == self.rank else [torch.empty_like(s) for s in src]
        for part in box:
          dist.broadcast(part, src=r, group=self.group)
        dst[r] = box[self.rank]
      recv.copy_(torch.cat(dst).to(recv.device))

  return HostStaged(native_comm=False)


def _shard(rows, lo, hi):
  seen = {}
  return [seen.setdefault(id(g), g[lo:hi].to(DEV)) for g in rows]


def _bulyan_mismatch_report(bm, agg, local, full, got, want, bad, lo, hi, kind, order_sharded):
  """What a Bulyan mismatch between the sharded and the unsharded call looks like (which side moved, where, whether it
  repeats).  NO collective in here: the peers may have passed and moved on (`order_sharded` is the ranking the sharded
  call used, a device tensor this rank already holds)."""
  idx = bad.nonzero().flatten()
  first, last = int(idx[0]), int(idx[-1])
  blocks = torch.unique(idx // 1024)
  m = N - F - 2
  got2 = agg.backend.bulyan_pass2(local, order_sharded, F, m)          # pass 2 of the shard again, same ranking
  want2 = bm.bulyan(full, F)                                          # (its ranking comes from the cache: pass 2 alone)
  order_single = bm.gars.bulyan_ranking(full, F)
  sharded_list = order_sharded[:N].tolist()
  cross = bm.gars.bulyan_pass2(full, order_sharded, F, m)             # the whole vectors with the SHARDED ranking
  host = O.bulyan([g.cpu() for g in full], F)[lo:hi].to(DEV)
  tol = 2e-6 * float(want.abs().max())

  def off(x):
    return int(((x - host).abs() > tol).sum())
  return (f"{kind} bulyan: {int(bad.sum())} coordinates of [{lo}, {hi}) differ (max {float((got - want[lo:hi]).abs().max()):.3e},"
          f" scale {float(want.abs().max()):.3e}); first {first}, last {last}, in {blocks.numel()} blocks of 1024 "
          f"({blocks[:8].tolist()}...); pass 2 of the shard again equal to its first run {bool(torch.equal(got2, got))}, "
          f"unsharded again equal to its first run {bool(torch.equal(want2, want))}; same ranking "
          f"{order_single == sharded_list} (single {order_single[:18]}, sharded {sharded_list[:18]}); coordinates off the "
          f"oracle on the host: sharded first run {off(got)}, unsharded first run {off(want[lo:hi])}, shard again "
          f"{off(got2)}, unsharded again {off(want2[lo:hi])}, whole vectors with the sharded ranking {off(cross[lo:hi])}")


def _step_mismatch_report(bm, single, got_def, want_def, off, lo, hi, gar, it):
  """A step whose sharded defense vector differs from the single-rank step's: where, by how much, and whether the
  single-rank side agrees with the rule applied to ITS OWN momentum buffers and Byzantine vector (no collective here:
  the peers may have passed and moved on)."""
  idx = off.nonzero().flatten()
  blocks = torch.unique(idx // 1024)
  rows = list(single.buffers) + [single.last_byzantine] * F
  rule = bm.krum if gar == "krum" else bm.bulyan
  again = rule(rows, F)
  order = (bm.gars.krum_selection(rows, F) if gar == "krum" else bm.gars.bulyan_ranking(rows, F))
  host = (O.krum if gar == "krum" else O.bulyan)([r.cpu() for r in rows], F).to(DEV)
  tol = 2e-6 * max(float(want_def.abs().max()), 1e-30)
  return (f"step {it}, rule {gar}: {int(off.sum())} coordinates of [{lo}, {hi}) differ (max "
          f"{float((got_def - want_def[lo:hi]).abs().max()):.3e}, scale {float(want_def.abs().max()):.3e}); first "
          f"{int(idx[0])}, last {int(idx[-1])}, in {blocks.numel()} blocks of 1024 ({blocks[:8].tolist()}...); the rule "
          f"on the single-rank step's own buffers: {int(((again - want_def).abs() > tol).sum())} coordinates off its "
          f"defense vector over the whole length, {int(((again[lo:hi] - got_def).abs() > tol).sum())} off the sharded "
          f"one on this slice; the oracle on the host on the same buffers: {int(((host - want_def).abs() > tol).sum())} "
          f"off the single-rank defense, {int(((host[lo:hi] - got_def).abs() > tol).sum())} off the sharded one; ranking "
          f"of the buffers {order[:20]}")


def _poison_allocator():
  """Fill what torch's caching allocator will hand out next with NaN bit patterns (0x7fc00000: NaN as a float, a
  huge count as an integer): a kernel that reads memory nobody wrote — an arrival counter assumed zero, a slot past
  the rows, an output block never stored — then fails every time instead of only when the memory happens to hold
  something else than a previous run's identical result."""
  big = torch.full((1 << 28,), math.nan, dtype=torch.float32, device=DEV)
  mid = [torch.full((1 << 14,), math.nan, dtype=torch.float32, device=DEV) for _ in range(512)]
  tiny = [torch.full((128,), math.nan, dtype=torch.float32, device=DEV) for _ in range(4096)]
  torch.cuda.synchronize()
  del big, mid, tiny


def _close(a, b, tol, what):
  scale = max(float(b.abs().max()) if b.numel() else 0.0, 1e-30)
  err = float((a - b).abs().max()) if b.numel() else 0.0
  assert err <= tol * scale, (what, err, scale)


def _worker(rank, world, rendezvous, d, queue):
  """Everything a rank does; whatever goes wrong travels to the parent as text (a bare exit code explains nothing).
  A rank that dies in native code (a signal, an abort of the HIP runtime, a GPU memory fault) raises nothing in
  Python: its stderr — the runtime's last words and faulthandler's dump of the Python stack at the signal — goes to a
  file next to the rendezvous store, which the parent quotes."""
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
  os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # the box's hostname may not resolve: loopback, explicitly
  # file rendezvous (no port to lose a race for); a short timeout: a rank that fails an assertion leaves its peers in a
  # collective, they must not wait for long
  dist.init_process_group("gloo", init_method=f"file://{rendezvous}", rank=rank, world_size=world,
                          timeout=datetime.timedelta(seconds=60))
  try:
    torch.cuda.set_device(0)
    if os.environ.get("BM_TEST_POISON", "1") != "0":
      _poison_allocator()
    import byzantinemomentum_amd as bm
    from byzantinemomentum_amd.sharded import HipBackend, ShardedAggregator, owned_workers, shard_bounds
    from byzantinemomentum_amd.step import AggregationStep
    agg = _staged_aggregator()
    assert isinstance(agg.backend, HipBackend) and agg.world_size == world and agg.collective and agg.native is None
    lo, hi = shard_bounds(d, world, rank)
    report = {"shard": (lo, hi)}
    light = d > 500000  # the long case is there for the plan of the distance pass: distances, Krum, Bulyan, one step
    for kind in (("hetero",) if light else ("hetero", "little")):
      rows, h = O.make_stack(kind, N, F, d, seed=1234)
      local = _shard(rows, lo, hi)
      full = _shard(rows, 0, d)
      assert agg.total_length(hi - lo) == d
      # the all-reduced squared distances against the single-GPU pass on the whole vectors
      sq = agg.global_sqdist(local)
      want_sq = bm.gars.pairwise_sqdist(full)
      off = ~torch.eye(N, dtype=torch.bool, device=DEV) & (want_sq > 0)
      assert float(((sq - want_sq).abs()[off] / want_sq[off]).max()) <= 1e-6, kind
      assert float(sq[h, h + 1]) == 0.0  # aliased Byzantine rows: exactly zero on every shard, hence in the sum
      # selections: identical to the unsharded call (the decisive gaps of these stacks are far above the 1e-6 above)
      kept = {}

      def sharded_bulyan():
        # agg.bulyan(local, F) in its pieces (what it runs without the library's own communicator), the ranking kept
        # for the report of a mismatch
        kept["order"] = agg.backend.rank(agg.global_sqdist(local), N, F, N - F - 2, bm._lib.RANK_BULYAN)
        return agg.backend.bulyan_pass2(local, kept["order"], F, N - F - 2)
      for name, sharded, single in (("krum", lambda: agg.krum(local, F), lambda: bm.krum(full, F)),
                                    ("krum m=1", lambda: agg.krum(local, F, 1), lambda: bm.krum(full, F, 1)),
                                    ("bulyan", lambda: sharded_bulyan(), lambda: bm.bulyan(full, F)),
                                    ("aksel", lambda: agg.aksel(local, F), lambda: bm.aksel(full, F)),
                                    ("cge", lambda: agg.cge(local, F), lambda: bm.cge(full, F)),
                                    ("brute", lambda: agg.brute(local, F), lambda: bm.brute(full, F))):
        if light and name in ("aksel", "cge", "brute"):
          continue
        got, want = sharded(), single()
        assert got.shape[0] == hi - lo
        if name == "bulyan":  # pass 2 may keep either of two exactly tied deviations: allow isolated columns only if tied
          # (`~(<=)`: a NaN — an output nobody stored, on poisoned memory — counts as a difference)
          bad = ~((got - want[lo:hi]).abs() <= 2e-6 * float(want.abs().max()))
          if int(bad.sum()) > max(1, (hi - lo) // 10000):
            raise AssertionError(_bulyan_mismatch_report(bm, agg, local, full, got, want, bad, lo, hi, kind, kept["order"]))
        else:
          assert torch.equal(got, want[lo:hi]), (kind, name)
        whole = agg.all_gather_output(got, d)
        assert whole.shape[0] == d and torch.equal(whole[lo:hi], got)
      # the coordinate-wise rules need no exchange: bit-identical slices
      assert torch.equal(agg.median(local), bm.median(full)[lo:hi])
      assert torch.equal(agg.trmean(local, F), bm.trmean(full, F)[lo:hi])
      assert torch.equal(agg.phocas(local, F), bm.phocas(full, F)[lo:hi])
      # statistics through the packed exchange
      avg, norm, dev, mx = agg.compute_avg_dev_max(local[:h])
      wavg, wnorm, wdev, wmx = bm.compute_avg_dev_max(full[:h])
      assert torch.equal(avg, wavg[lo:hi])
      assert abs(norm - wnorm) <= 1e-9 * wnorm and abs(dev - wdev) <= 1e-9 * wdev and mx == wmx
      report[kind] = (norm, dev, mx)
    # worker-parallel production -> dimension-major, GPU tensors through the (staged) all-to-all
    if not light:
      rows, _ = O.make_stack("iid", N, 0, d, seed=77)
      mine = [rows[i].to(DEV) for i in owned_workers(N, world, rank)]
      got = agg.to_dim_sharded(mine, N, d, device=torch.device(DEV))
      assert len(got) == N
      for i in range(N):
        assert got[i].device.type == "cuda" and torch.equal(got[i].cpu(), rows[i][lo:hi]), i
      if hi > lo:
        assert torch.equal(agg.median(got), bm.median(_shard(rows, 0, d))[lo:hi])  # views of the receive buffer feed the kernels
    # the full step (worker momentum, empire, study block) on the slice against the single-rank step on the whole vectors
    for gar in (("krum", "bulyan", "median") if d <= 500000 else ("krum",)):  # (the long case: one rule, the point is the plan)
      sharded = AggregationStep(N, F, F, gar=gar, momentum=0.9, dampening=0.9, attack_factor=1.1, nb_past=2, aggregator=agg)
      single = AggregationStep(N, F, F, gar=gar, momentum=0.9, dampening=0.9, attack_factor=1.1, nb_past=2,
                               aggregator=ShardedAggregator(local_only=True))  # (the default one would span the job)
      gen = torch.Generator().manual_seed(5)
      origin = torch.randn(d, generator=gen)
      params = origin + 0.01
      for it in range(4):
        base = 0.2 * torch.randn(d, generator=gen)
        sampled = [base + (0.5 + 0.05 * i) * torch.randn(d, generator=gen) for i in range(N - F)]
        got_def = sharded.run([g[lo:hi].to(DEV) for g in sampled], params[lo:hi].to(DEV), origin[lo:hi].to(DEV))
        got = sharded.floats()
        want_def = single.run([g.to(DEV) for g in sampled], params.to(DEV), origin.to(DEV))
        want = single.floats()
        if gar == "median":
          assert torch.equal(got_def, want_def[lo:hi]), (gar, it)
        else:
          scale = max(float(want_def.abs().max()), 1e-30)
          off = (got_def - want_def[lo:hi]).abs() > 2e-6 * scale
          if bool(off.any()):
            raise AssertionError(_step_mismatch_report(bm, single, got_def, want_def, off, lo, hi, gar, it))
        for key, val in want.items():
          g = got[key]
          assert (math.isnan(g) and math.isnan(val)) or abs(g - val) <= 1e-6 * max(abs(val), 1e-6), (gar, it, key, g, val)
        report[(gar, it)] = tuple(sorted((k, v) for k, v in got.items() if not math.isnan(v)))
    # the same step with the attack's factor search (attacks/identical.py:67-77, the reference's default factor=-16): the
    # distance pass over honests + [avg, avg + att] is all-reduced, then every rank searches on the device from the same
    # (h+2)^2 scalars — the same candidates as the single-rank step, the same factor on every shard
    if d <= 500000:
      sharded = AggregationStep(N, F, F, gar="krum", momentum=0.9, dampening=0.9, nb_past=2, attack_evals=8, aggregator=agg)
      single = AggregationStep(N, F, F, gar="krum", momentum=0.9, dampening=0.9, nb_past=2, attack_evals=8,
                               aggregator=ShardedAggregator(local_only=True))
      gen = torch.Generator().manual_seed(6)
      origin = torch.randn(d, generator=gen)
      for it in range(3):
        base = 0.2 * torch.randn(d, generator=gen)
        sampled = [base + (0.5 + 0.05 * i) * torch.randn(d, generator=gen) for i in range(N - F)]
        got_def = sharded.run([g[lo:hi].to(DEV) for g in sampled], origin[lo:hi].to(DEV), origin[lo:hi].to(DEV))
        want_def = single.run([g.to(DEV) for g in sampled], origin.to(DEV), origin.to(DEV))
        assert isinstance(sharded._factor_now, torch.Tensor), "the sharded step must search on the device as well"
        (got_f, got_tr), (want_f, want_tr) = (sharded.last_factor, sharded.last_search), (single.last_factor, single.last_search)
        assert [x for x, _ in got_tr] == [x for x, _ in want_tr] and got_f == want_f, (it, got_tr, want_tr)
        for (_, y), (_, yo) in zip(got_tr, want_tr):
          assert abs(y - yo) <= 1e-6 * max(abs(yo), 1e-12), (it, y, yo)
        scale = max(float(want_def.abs().max()), 1e-30)
        if hi > lo:  # (an empty shard takes part in the exchange and the search, and has nothing to compare)
          assert float((got_def - want_def[lo:hi]).abs().max()) <= 2e-6 * scale, it
        report[("krum-search", it)] = (got_f, tuple(x for x, _ in got_tr))
    torch.cuda.synchronize()
    queue.put((rank, report))
    dist.barrier()
  finally:
    dist.destroy_process_group()


def _last_words(rendezvous, world, limit=4000):
  """The tail of every rank's stderr file (empty files are left out)."""
  parts = []
  for r in range(world):
    try:
      with open(os.path.join(os.path.dirname(rendezvous), f"rank{r}.stderr")) as fh:
        text = fh.read().strip()
    except OSError:
      continue
    if text:
      parts.append(f"--- stderr of rank {r} ---\n{text[-limit:]}")
  return "\n".join(parts)


def _run_ranks(world, d):
  """One attempt: spawn the ranks, collect their reports.  Returns (reports by rank, exit codes, the ranks' stderr)."""
  import tempfile
  import time
  time.sleep(2.0)  # (let the processes of a previous case finish tearing down their device contexts)
  ctx = mp.get_context("spawn")
  queue = ctx.Queue()
  rendezvous = os.path.join(tempfile.mkdtemp(prefix="bm_multirank_"), "store")
  procs = [ctx.Process(target=_worker, args=(r, world, rendezvous, d, queue)) for r in range(world)]
  for p in procs:
    p.start()
  import queue as queue_mod
  results = {}
  deadline = time.time() + 800
  try:
    while len(results) < world and time.time() < deadline:
      try:
        rank, rep = queue.get(timeout=5)
        results[rank] = rep
      except queue_mod.Empty:
        if any(p.exitcode not in (None, 0) for p in procs):
          break  # a rank failed: its peers sit in a collective, do not wait for them
  finally:
    for p in procs:
      p.join(timeout=30 if len(results) == world else 1)
      if p.is_alive():
        p.terminate()
        p.join(timeout=10)
  return results, [p.exitcode for p in procs], _last_words(rendezvous, world)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,d", [(2, 200003), (3, 130), (4, 300), (4, 1 << 20)])
def test_multi_rank_sharded_path_on_the_hip_kernels(world, d):
  """world 2, d = 200 003: two long ragged shards (the second one is not a multiple of 4 coordinates long: the
  kernels' tail paths); world 3, d = 130: shards of 64, 64 and 2 coordinates; world 4, d = 300: 128, 128, 44 and an
  EMPTY one; world 4, d = 2^20: the length at which the distance pass changes its split plan — every rank must plan
  from the total, not from its 262 144 coordinates."""
  results, codes, words = _run_ranks(world, d)
  errors = {r: rep["error"] for r, rep in results.items() if "error" in rep}
  # (exit codes: a negative one is the signal that killed the rank; the stderr files hold what Python never saw.  One
  #  attempt: a mismatch is a parity failure, a rank that died is a failure with its last words — nothing is repeated.)
  report = (f"exit codes {codes}\n" + "\n".join(f"--- rank {r} ---\n{text}" for r, text in sorted(errors.items())) + "\n" + words)
  assert not errors and len(results) == world and all(c == 0 for c in codes), report
  # every rank decoded the same floats from the same packed exchange
  keys = [k for k in results[0] if k != "shard"]
  for r in range(1, world):
    for k in keys:
      assert results[r][k] == results[0][k], (r, k)
