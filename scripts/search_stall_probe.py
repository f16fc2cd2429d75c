"""The bench's first series of factor searches has ONE call of 37-130 ms among ten, at the 8th-9th call, in every run of
rounds 5-6 — whichever form runs first — and no probe outside the bench shows it.  This probe replays the bench's own
sequence in front of the series (extras_single_gpu: the two timed loops of krum_c3) and splits every search into the time
to queue it and the time to wait for it, with the allocator's and the garbage collector's counters beside them."""
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
    events.append((info["generation"], round((time.perf_counter() - state["t0"]) * 1e3, 2)))


gc.callbacks.append(on_gc)
dev = torch.device("cuda:0")
n, f, d = 51, 12, bench.D_RESNET18
bench.SEPARATE_ROWS = True
timer = bench.KernelTimer()
stacks = bench.make_stacks(n, f, d, dev, 2, 4321, False)
bench.timed_loop(lambda i: bm.krum(stacks[i & 1], f), 12, 3, timer, "krum_c3")
bench.timed_loop(lambda i: bm.gars.pairwise_sqdist(stacks[i & 1]), 12, 3, timer, "krum_c3_dist")
honests = stacks[0][:n - f]
avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
for mode in (sys.argv[1:] or ["auto", "host", "auto"]):
  runner = AggregationStep(n, f, f, gar="krum", attack_evals=16, line_search=mode, nb_past=0)
  runner._search_factor(honests, avg, direction)
  torch.cuda.synchronize()
  rows = []
  for i in range(20):
    del events[:]
    before = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    runner.last_factor = runner._search_factor(honests, avg, direction)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    after = torch.cuda.memory_stats(dev)
    rows.append((i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, after["num_device_alloc"] - before["num_device_alloc"],
                 after["num_device_free"] - before["num_device_free"], after["num_alloc_retries"] - before["num_alloc_retries"], list(events)))
  med = sorted(r[1] + r[2] for r in rows)[10]
  print(f"line_search={mode}: median {med:.3f} ms; calls over twice the median (index, queue ms, wait ms, device allocs, device frees, "
        f"retries, gc): {[(r[0], round(r[1], 2), round(r[2], 2)) + r[3:] for r in rows if r[1] + r[2] > 2 * med]}", flush=True)
  print("   queue ms: " + " ".join(f"{r[1]:.2f}" for r in rows), flush=True)
  print("   wait  ms: " + " ".join(f"{r[2]:.2f}" for r in rows), flush=True)
