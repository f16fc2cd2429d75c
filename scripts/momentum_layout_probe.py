"""Does the PLACEMENT of the 40 + 3 row buffers of bm_momentum_stats (C5 shape: 20 sampled gradients, 20 momentum
buffers, d = 36 546 980) change its time?  Rows cut out of one slab at different strides against separately allocated
tensors (what torch's caching allocator hands out depends on what ran before).  One process, HIP events, queue kept full.

    python scripts/momentum_layout_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402

D = 36546980
H = 20
MB2 = 2 << 20


def timed(g_rows, b_rows, reps=8, rounds=3):
  for _ in range(2):
    bm.stats.momentum_stats(g_rows, b_rows, 0.99, 0.01, None, 1.1, "empire")
  torch.cuda.synchronize()
  best = []
  for _ in range(rounds):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
      bm.stats.momentum_stats(g_rows, b_rows, 0.99, 0.01, None, 1.1, "empire")
    b.record()
    torch.cuda.synchronize()
    best.append(a.elapsed_time(b) * 1e3 / reps)
  best.sort()
  return best[len(best) // 2], best[0]


def slab_rows(count, stride_bytes, dev):
  stride = stride_bytes // 4
  slab = torch.empty(stride * count + 64, dtype=torch.float32, device=dev)
  base = (-slab.data_ptr() % 256) // 4
  rows = [slab[base + i * stride: base + i * stride + D] for i in range(count)]
  for r in rows:
    r.normal_()
  return slab, rows


def main():
  dev = torch.device("cuda:0")
  nbytes = 9209838960
  up = lambda x, a: (x + a - 1) // a * a  # noqa: E731
  row = D * 4
  layouts = [("separate torch.empty per row", None),
             ("slab, rows packed (stride = d*4 up to 256 B)", up(row, 256)),
             ("slab, stride = multiple of 2 MB", up(row, MB2)),
             ("slab, stride = 2 MB multiple + 4 KB + 256 B", up(row, MB2) + 4096 + 256),
             ("slab, stride = 2 MB multiple + 64 KB + 256 B", up(row, MB2) + 65536 + 256),
             ("slab, stride = 2 MB multiple + 1 MB + 4 KB", up(row, MB2) + (1 << 20) + 4096),
             ("separate torch.empty per row (again)", None)]
  for name, stride in layouts:
    if stride is None:
      keep = None
      g_rows = [torch.randn(D, device=dev) for _ in range(H)]
      b_rows = [torch.randn(D, device=dev) for _ in range(H)]
    else:
      keep, rows = slab_rows(2 * H, stride, dev)
      g_rows, b_rows = rows[:H], rows[H:]
    med, best = timed(g_rows, b_rows)
    span = sorted(r.data_ptr() % MB2 for r in g_rows + b_rows)
    print(f"{name:52s}: {med:8.1f} us (best {best:8.1f}) = {nbytes / med / 1e3:5.0f} GB/s   "
          f"row offsets mod 2 MB: {span[0]} .. {span[-1]}")
    del g_rows, b_rows, keep
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()
