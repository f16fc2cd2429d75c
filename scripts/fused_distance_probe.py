"""First pass of a C5 step with the distance pass riding along (bm_momentum_stats_sqdist) against the two passes one
after the other, HIP events, queue kept full."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402

D, H, NB = 36546980, 20, 5
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
sampled = [torch.randn(D, device=dev, generator=gen) for _ in range(H)]
bufs = [0.1 * torch.randn(D, device=dev, generator=gen) for _ in range(H)]


def timed(fn, reps=8, rounds=3):
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
  return sorted(res)[len(res) // 2]


def fused():
  return bm.stats.momentum_stats_sqdist(sampled, bufs, 0.99, 0.01, None, 1.1, "empire", NB)


def separate():
  s, h, z, o = bm.stats.momentum_stats(sampled, bufs, 0.99, 0.01, None, 1.1, "empire")
  return bm.gars.pairwise_sqdist(bufs + [z] * NB)


for _ in range(2):
  print(f"fused {timed(fused):8.1f} us   separate {timed(separate):8.1f} us")
