"""Does the PLACEMENT of the row buffers change the time of the multi-row kernels?  Rows cut out of one slab at
strides = (multiple of 2 MB) + skew, against separately allocated tensors (what torch's caching allocator hands out
depends on what ran before).  Median / trimmed mean (25 rows), pairwise distances (51 rows), first pass of a step
(20 + 20 rows).  One process, HIP events around back-to-back calls.

    python scripts/layout_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402

MB2 = 2 << 20
DEV = torch.device("cuda:0")


def timed(fn, reps=10, rounds=3):
  for _ in range(2):
    fn()
  torch.cuda.synchronize()
  res = []
  for _ in range(rounds):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
      fn()
    b.record()
    torch.cuda.synchronize()
    res.append(a.elapsed_time(b) * 1e3 / reps)
  res.sort()
  return res[len(res) // 2]


def rows_of(count, d, skew):
  """skew None: separate allocations; else one slab, stride = d*4 rounded up to 2 MB, plus skew bytes."""
  if skew is None:
    return None, [torch.randn(d, device=DEV) for _ in range(count)]
  stride = ((d * 4 + MB2 - 1) // MB2 * MB2 + skew) // 4
  slab = torch.empty(stride * count + 64, dtype=torch.float32, device=DEV)
  base = (-slab.data_ptr() % 256) // 4
  rows = [slab[base + i * stride: base + i * stride + d] for i in range(count)]
  for r in rows:
    r.normal_()
  return slab, rows


def main():
  skews = [None, 0, 256, 1280, 4352, 8448, 69888, (1 << 20) + 4352, None]
  d2, d5 = 11173962, 36546980
  print("layout".ljust(28) + "median25   trmean25   pairwise51   momentum20+20   (us)")
  for skew in skews:
    keep1, st25 = rows_of(25, d2, skew)
    t_med = timed(lambda: bm.median(st25))
    t_trm = timed(lambda: bm.trmean(st25, 5))
    del st25, keep1
    torch.cuda.empty_cache()
    keep2, st51 = rows_of(51, d2, skew)
    t_pair = timed(lambda: bm.gars.pairwise_sqdist(st51))
    del st51, keep2
    torch.cuda.empty_cache()
    keep3, st40 = rows_of(40, d5, skew)
    t_mom = timed(lambda: bm.stats.momentum_stats(st40[:20], st40[20:], 0.99, 0.01, None, 1.1, "empire"), reps=6)
    del st40, keep3
    torch.cuda.empty_cache()
    name = "separate allocations" if skew is None else f"slab, 2 MB multiple + {skew}"
    print(f"{name:28s}{t_med:8.1f}   {t_trm:8.1f}   {t_pair:9.1f}   {t_mom:12.1f}", flush=True)


if __name__ == "__main__":
  main()
