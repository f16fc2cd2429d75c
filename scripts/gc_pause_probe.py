"""Are the 40-130 ms stalls that hit one call in ten or twenty of the bench's short timed loops (rounds 4-6: "a loaded
host") Python's own cyclic garbage collector?  A full (generation 2) collection walks every tracked object of the
process — millions once torch is imported.  This probe logs every collection (gc.callbacks: generation, duration) while
it times 60 factor searches one by one, with the collector enabled and then disabled."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import byzantinemomentum_amd as bm  # noqa: E402
from byzantinemomentum_amd.step import AggregationStep  # noqa: E402

events = []
state = {}


def on_gc(phase, info):
  if phase == "start":
    state["t0"] = time.perf_counter()
  else:
    events.append((info["generation"], (time.perf_counter() - state["t0"]) * 1e3, info["collected"]))


gc.callbacks.append(on_gc)
dev = torch.device("cuda:0")
n, f, d = 51, 12, bench.D_RESNET18
bench.SEPARATE_ROWS = True
stacks = bench.make_stacks(n, f, d, dev, 2, 4321, False)
timer = bench.Timer(dev) if hasattr(bench, "Timer") else None
honests = stacks[0][:n - f]
avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
print(f"tracked objects: {len(gc.get_objects())}, thresholds {gc.get_threshold()}, counts {gc.get_count()}", flush=True)
t0 = time.perf_counter()
gc.collect()
print(f"one full gc.collect(): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
for label, enabled in (("collector enabled", True), ("collector disabled", False), ("collector enabled", True)):
  gc.enable() if enabled else gc.disable()
  runner = AggregationStep(n, f, f, gar="krum", attack_evals=16, line_search="auto", nb_past=0)
  runner._search_factor(honests, avg, direction)
  torch.cuda.synchronize()
  del events[:]
  each = []
  for _ in range(60):
    t0 = time.perf_counter()
    runner.last_factor = runner._search_factor(honests, avg, direction)
    torch.cuda.synchronize()
    each.append((time.perf_counter() - t0) * 1e3)
  slow = [(i, round(v, 2)) for i, v in enumerate(each) if v > 2 * sorted(each)[30]]
  print(f"{label}: median {sorted(each)[30]:.3f} ms, mean {sum(each) / 60:.3f} ms, calls over twice the median: {slow}; "
        f"collections during the loop (generation, ms, collected): {[(g, round(ms, 2), c) for g, ms, c in events]}", flush=True)
gc.enable()
