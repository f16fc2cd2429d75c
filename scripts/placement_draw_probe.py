"""Which placement of 51 separately allocated rows does a process draw, and what does the distance pass make of it?
Twelve times: the allocator's cache emptied, bench.make_stacks(51, 12) (two stacks, one torch.empty per row, as bench.py's
krum_c3 entry allocates them), then per STACK: the distance pass (bm_pairwise_sqdist) and Multi-Krum timed with HIP events
(median of 10 calls, ranking cache emptied), next to what the row addresses look like — how many distinct offsets modulo
2 MB the 51 rows have, and the smallest / largest gap between consecutive rows.  Every other draw is preceded by a 1.7 GB
block that is allocated and released (what torch.stack leaves in the cache), to see both allocator histories."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import byzantinemomentum_amd as bm  # noqa: E402
from byzantinemomentum_amd import gars  # noqa: E402

dev = torch.device("cuda:0")
n, f, d = 51, 12, bench.D_RESNET18
bench.SEPARATE_ROWS = True


def timed(fn, reps=10):
  each = []
  for _ in range(3):
    gars.invalidate_rank_cache()
    fn()
  for _ in range(reps):
    gars.invalidate_rank_cache()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    each.append(a.elapsed_time(b))
  return sorted(each)[len(each) // 2]


for draw in range(12):
  torch.cuda.empty_cache()
  if draw % 2 == 1:
    big = torch.empty(39 * d, device=dev)
    del big
  stacks = bench.make_stacks(n, f, d, dev, 2, 4321 + draw, False)
  for s, rows in enumerate(stacks):
    ptrs = [r.data_ptr() for r in rows]
    mod = sorted({p % (2 << 20) for p in ptrs})
    order = sorted(ptrs)
    gaps = [b - a for a, b in zip(order, order[1:])]
    t_dist = timed(lambda: gars.pairwise_sqdist(rows))
    t_krum = timed(lambda: bm.krum(rows, f))
    print(f"draw {draw:2d} stack {s}: distance pass {t_dist * 1e3:6.1f} us  krum {t_krum * 1e3:6.1f} us  | distinct offsets mod 2 MB {len(mod):2d}"
          f"  gaps min {min(gaps) / (1 << 20):8.3f} MB max {max(gaps) / (1 << 20):9.3f} MB  in address order {ptrs == order}", flush=True)
  del stacks, rows
