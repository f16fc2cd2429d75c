"""Brute (aggregators/brute.py:32-80) end to end at n = 25, f = 5 and n = 51, f = 12, d = 11 173 962, for a kernel
trace of the FINAL form of the search (round 5's review: "no kernel trace of the final form was taken"):
  rocprofv3 --kernel-trace --stats -- python scripts/brute_probe.py
prints the wall clock per call; the trace says where it goes (distance pass, reduction, brute_select_kernel, mean)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import byzantinemomentum_amd as bm  # noqa: E402
from byzantinemomentum_amd import gars  # noqa: E402

dev = torch.device("cuda:0")
bench.SEPARATE_ROWS = True
for n, f in ((25, 5), (51, 12)):
  d = bench.D_RESNET18
  stacks = bench.make_stacks(n, f, d, dev, 2, 1234 + n, False)
  for i in range(3):
    gars.invalidate_rank_cache()
    out = bm.brute(stacks[i & 1], f)
  torch.cuda.synchronize()
  each = []
  for i in range(12):
    gars.invalidate_rank_cache()
    t0 = time.perf_counter()
    out = bm.brute(stacks[i & 1], f)
    torch.cuda.synchronize()
    each.append((time.perf_counter() - t0) * 1e3)
  status = int(out.brute_status.item())
  print(f"brute n={n} f={f} d={d}: median {sorted(each)[len(each) // 2]:.3f} ms, min {min(each):.3f} ms per call (wall clock, "
        f"synchronised per call); status {status}; checksum {float(out.double().sum()):.9e}", flush=True)
  del stacks
