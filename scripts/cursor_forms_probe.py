"""The factor search against the median, the trimmed mean and MeaMed (C2 shape: n = 25, f = 5, d = 11 173 962) with the
exploration's cursor in device memory (line_search="auto": bm_search_device_next, the host queues the sixteen evaluations
and waits once) and on the host ("host": one synchronisation per evaluation) — same kernels, same candidates.  Alternating,
ten searches each, wall clock per search, synchronised after every search."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import byzantinemomentum_amd as bm  # noqa: E402
from byzantinemomentum_amd.step import AggregationStep  # noqa: E402

dev = torch.device("cuda:0")
n, f, d = 25, 5, bench.D_RESNET18
bench.SEPARATE_ROWS = True
stacks = bench.make_stacks(n, f, d, dev, 1, 4321, False)
honests = stacks[0][:n - f]
avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
for gar in ("median", "trmean", "meamed", "aksel"):
  runners = {mode: AggregationStep(n, f, f, gar=gar, attack_evals=16, line_search=mode, nb_past=0) for mode in ("auto", "host")}
  times = {mode: [] for mode in runners}
  for rep in range(12):
    for mode, runner in runners.items():
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      runner.last_factor = runner._search_factor(honests, avg, direction)
      t1 = time.perf_counter()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      if rep >= 2:
        times[mode].append(((t2 - t0) * 1e3, (t1 - t0) * 1e3))
  assert runners["auto"].last_factor == runners["host"].last_factor and runners["auto"].last_search == runners["host"].last_search
  line = f"{gar:7s}"
  for mode, each in times.items():
    total = sorted(v[0] for v in each)[len(each) // 2]
    queued = sorted(v[1] for v in each)[len(each) // 2]
    line += f"  cursor on the {'device' if mode == 'auto' else 'host  '}: {total:.3f} ms per search (host busy {queued:.3f} ms)"
  print(line + f"  factor {runners['auto'].last_factor}", flush=True)
