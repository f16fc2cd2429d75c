"""Per-rank cost of the dim-sharded path at P = 8, measured on ONE GPU: a one-rank RCCL group with forced
collectives runs exactly the code a rank of an 8-GPU job runs, on 1/8 of the coordinates.  Reports wall time
per call with the queue kept full (what a training loop sees) for the single-call C entry points against the
call-by-call Python sequence with torch.distributed collectives.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 scripts/per_rank_probe.py
"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402
from byzantinemomentum_amd.sharded import ShardedAggregator, shard_bounds  # noqa: E402
from byzantinemomentum_amd.step import AggregationStep  # noqa: E402


def wall_us(fn, reps=200, warm=20):
  for i in range(warm):
    fn(i)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(reps):
    fn(i)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e6


def main():
  device = torch.device("cuda", 0)
  torch.cuda.set_device(device)
  dist.init_process_group("nccl", device_id=device)
  P = 8
  n, f = 25, 5
  gen = torch.Generator(device=device).manual_seed(3)
  for name, d_total in (("C4 bulyan", 11173962), ("C5 step (krum)", 36546980), ("C5 step (median)", 36546980)):
    lo, hi = shard_bounds(d_total, P, 0)
    d = hi - lo
    mu = 0.1 * torch.randn(d, device=device, generator=gen)
    stacks = [[mu + s * torch.randn(d, device=device, generator=gen) for s in torch.linspace(0.5, 1.5, n).tolist()]
              for _ in range(2)]
    native = ShardedAggregator(force_collectives=True)
    plain = ShardedAggregator(force_collectives=True, native_comm=False)
    if name.startswith("C4"):
      a = wall_us(lambda i: native.bulyan(stacks[i & 1], f))
      b = wall_us(lambda i: plain.bulyan(stacks[i & 1], f))
      ev = []
      for i in range(50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); native.bulyan(stacks[i & 1], f); e1.record(); ev.append((e0, e1))
      torch.cuda.synchronize()
      gpu = sum(x.elapsed_time(y) for x, y in ev) / len(ev) * 1e3
      print(f"{name}: d_local={d} (1/{P} of {d_total}); one C call {a:.0f} us/agg wall, Python sequence + torch.distributed "
            f"{b:.0f} us/agg wall; GPU time of the single call {gpu:.0f} us (HIP events)", flush=True)
      # the same single call recorded into a HIP graph (byzantinemomentum_amd/graphs.py), one graph per stack
      from byzantinemomentum_amd.graphs import GraphedCall
      try:
        graphs = [GraphedCall(lambda s=s: native.bulyan(stacks[s], f)) for s in (0, 1)]
        same = all(torch.equal(graphs[s](), native.bulyan(stacks[s], f)) for s in (0, 1))
        c = wall_us(lambda i: graphs[i & 1]())
        ev = []
        for i in range(50):
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record(); graphs[i & 1](); e1.record(); ev.append((e0, e1))
        torch.cuda.synchronize()
        gpu_g = sum(x.elapsed_time(y) for x, y in ev) / len(ev) * 1e3
        print(f"{name}: HIP graph replay of the single call {c:.0f} us/agg wall, {gpu_g:.0f} us GPU time (HIP events); "
              f"same bits as the eager call: {same}", flush=True)
        # and without any collective (what the launches alone cost): the unsharded rule on the same shard
        eager = wall_us(lambda i: bm.bulyan(stacks[i & 1], f))
        g1 = [GraphedCall(lambda s=s: bm.bulyan(stacks[s], f)) for s in (0, 1)]
        rep = wall_us(lambda i: g1[i & 1]())
        print(f"{name}: no collective: eager {eager:.0f} us/agg wall, graph replay {rep:.0f} us/agg wall", flush=True)
      except Exception as err:  # noqa: BLE001
        print(f"{name}: HIP graph capture failed: {err!r}", flush=True)
      if "--only-c4" in sys.argv:
        break
    else:
      gar = "krum" if "krum" in name else "median"
      h = n - f
      res = {}
      for label, agg, single in (("one C call", native, True), ("Python sequence", plain, False)):
        step = AggregationStep(n, f, f, gar=gar, momentum=0.99, dampening=0.99, nb_past=25, aggregator=agg,
                               single_call=single)

        def one(i):
          step.run(stacks[i & 1][:h])
          step.floats()
        res[label] = wall_us(one, reps=100, warm=30)
      print(f"{name}: d_local={d}; " + "; ".join(f"{k} {v:.0f} us/step wall (incl. the floats() sync)" for k, v in res.items()),
            flush=True)
    del stacks
  dist.destroy_process_group()


if __name__ == "__main__":
  main()
