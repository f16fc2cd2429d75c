"""A C5-size step with the momentum at the update (the reference's default placement), rule median and krum, 30 steps each
after 27 warm-up steps (the deque of 25 past averages is full) — the command a kernel trace is taken of:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o trace -- python scripts/step_update_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from byzantinemomentum_amd.step import AggregationStep  # noqa: E402


def main():
  dev = torch.device("cuda:0")
  n, f, d = 25, 5, 36546980
  h = n - f
  gen = torch.Generator(device=dev).manual_seed(77)
  mu_vec = 0.1 * torch.randn(d, device=dev, generator=gen)
  sets = [[mu_vec + s * torch.randn(d, device=dev, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()] for _ in range(2)]
  for gar in ("median", "krum"):
    runner = AggregationStep(n, f, f, gar=gar, momentum=0.99, dampening=0.99, momentum_at="update", attack_factor=1.1, nb_past=25)
    for i in range(27):
      runner.run(sets[i & 1])
      runner.floats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
      runner.run(sets[i & 1])
      runner.floats()
    torch.cuda.synchronize()
    print(f"update placement, rule {gar}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms per step (wall, floats() every step)", flush=True)


if __name__ == "__main__":
  main()
