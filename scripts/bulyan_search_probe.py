"""The factor search against Bulyan (attacks/identical.py:67-77, 16 evaluations) with the evaluate-only second pass
(bm_bulyan_pass2_eval: candidate in registers, objective in the same kernel) and with the written form it replaces
(bm_multi_fma3 + bm_bulyan_pass2 + bm_sqdist2), both with the cursor and the ranking on the device (line_search="auto"), and
the evaluate-only form with cursor and ranking on the host (line_search="host") — same candidates.  C4 shape (n = 25, f = 5) and the
reference's largest (n = 51, f = 12) at d = 11 173 962, the two forms alternating, wall clock per search, synchronised
after every search; then one evaluation of each form under HIP events."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import byzantinemomentum_amd as bm  # noqa: E402
from byzantinemomentum_amd import _lib, gars  # noqa: E402
from byzantinemomentum_amd.step import AggregationStep  # noqa: E402

dev = torch.device("cuda:0")
d = int(os.environ.get("D", bench.D_RESNET18))
bench.SEPARATE_ROWS = True
for n, f in ((25, 5), (51, 12)):
  stacks = bench.make_stacks(n, f, d, dev, 1, 4321, False)
  honests = stacks[0][:n - f]
  avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
  runner = AggregationStep(n, f, f, gar="bulyan", attack_evals=16, nb_past=0)
  supported = runner.ops.bulyan_pass2_eval_supported
  forms = {"evaluate-only": supported, "written": lambda *a, **k: False}
  times = {name: [] for name in forms}
  found = {}
  for rep in range(10):
    for name, fn in forms.items():
      runner.ops.bulyan_pass2_eval_supported = fn
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      runner.last_factor = runner._search_factor(honests, avg, direction)
      torch.cuda.synchronize()
      if rep >= 2:
        times[name].append((time.perf_counter() - t0) * 1e3)
      found[name] = (runner.last_factor, list(runner.last_search))
  runner.ops.bulyan_pass2_eval_supported = supported
  runner.last_factor = found["evaluate-only"][0]
  assert found["evaluate-only"][0] == found["written"][0], found
  worst = max(abs(a[1] - b[1]) / abs(b[1]) for a, b in zip(found["evaluate-only"][1], found["written"][1]))
  # the same search with the cursor and the ranking on the host (line_search="host": one synchronisation per evaluation)
  hosted = AggregationStep(n, f, f, gar="bulyan", attack_evals=16, nb_past=0, line_search="host")
  times["cursor and ranking on the host (evaluate-only)"] = []
  for rep in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    found["host"] = (hosted._search_factor(honests, avg, direction), list(hosted.last_search))
    torch.cuda.synchronize()
    if rep >= 2:
      times["cursor and ranking on the host (evaluate-only)"].append((time.perf_counter() - t0) * 1e3)
  assert found["host"][0] == runner.last_factor and found["host"][1] == found["evaluate-only"][1], (found["host"], found["evaluate-only"])
  line = f"n={n} f={f} d={d}"
  for name, each in times.items():
    line += f"  {name}: {sorted(each)[len(each) // 2]:.3f} ms per search"
  print(line + f"  factor {found['written'][0]}  largest relative difference of an objective {worst:.1e}", flush=True)
  # one evaluation of each form, HIP events, 20 times
  m = n - f - 2
  cand = torch.empty_like(avg)
  bm.stats.multi_fma3([cand], [avg], [direction], 1.0, 1.1)
  rows = honests + [cand] * f
  order, _ = gars._rank(rows, f, m, _lib.RANK_BULYAN)

  def written():
    bm.stats.multi_fma3([cand], [avg], [direction], 1.0, 1.1)
    return bm.stats.sqdist2(gars.bulyan_pass2(rows, order, f, m), avg)

  def fused():
    return bm.stats.bulyan_pass2_eval(honests, f, order, f, m, avg, direction, 1.1)

  for name, fn in (("evaluate-only", fused), ("written", written), ("evaluate-only", fused), ("written", written)):
    for _ in range(3):
      fn()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(20):
      fn()
    stop.record()
    torch.cuda.synchronize()
    print(f"    one evaluation, {name}: {start.elapsed_time(stop) / 20 * 1e3:.1f} us", flush=True)
  del stacks, honests, rows
