"""Time bm_momentum_stats alone at C5 (20 sampled gradients + 20 momentum buffers, d = 36 546 980); run once per value
of an environment knob (BM_STEP_BLOCKS, BM_STEP_STREAM, ...) for A/B comparisons inside one gpurun call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402


def main():
  dev = torch.device("cuda:0")
  h, d = 20, 36546980
  gen = torch.Generator(device=dev).manual_seed(5)
  sets = [[torch.randn(d, device=dev, generator=gen) for _ in range(h)] for _ in range(2)]
  bufs = [torch.zeros(d, device=dev) for _ in range(h)]
  for i in range(3):
    out = bm.stats.momentum_stats(sets[i & 1], bufs, 0.99, 0.01, None, 1.1, "empire")
  torch.cuda.synchronize()
  reps, rounds = 10, 5
  us = []
  for _ in range(rounds):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
      out = bm.stats.momentum_stats(sets[i & 1], bufs, 0.99, 0.01, None, 1.1, "empire")
    b.record()
    torch.cuda.synchronize()
    us.append(a.elapsed_time(b) * 1e3 / reps)
  us.sort()
  nbytes = 4 * d * (3 * h + 3)
  knobs = {k: v for k, v in os.environ.items() if k.startswith("BM_")}
  print(f"{knobs}: momentum_stats {us[rounds // 2]:.1f} us per call (best round {us[0]:.1f}) = "
        f"{nbytes / us[rounds // 2] / 1e3:.0f} GB/s; out6 {[round(v, 3) for v in out[3].tolist()[:2]]}")


if __name__ == "__main__":
  main()
