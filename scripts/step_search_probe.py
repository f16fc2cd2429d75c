"""A whole step WITH the factor search (the reference's default attack: factor=-16, attacks/identical.py:67-77) against
Multi-Krum, n = 25, f = 5, momentum at the update (the reference's default placement): the search evaluated on the
device (line_search="auto": the factor never leaves the GPU, the host queues the whole step ahead and waits once, in
floats()) next to the search evaluated on the host ("host": a copy and a stream synchronisation in the middle of the
step, then the rest of the step is launched into an idle queue).  The two runners alternate step by step on the same
gradients; median and mean of the wall clock per step, floats() every step like the reference's study file."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from byzantinemomentum_amd.step import AggregationStep  # noqa: E402


def main():
  dev = torch.device("cuda:0")
  n, f = 25, 5
  d = int(os.environ.get("D", "11173962"))
  h = n - f
  gen = torch.Generator(device=dev).manual_seed(78)
  mu_vec = 0.1 * torch.randn(d, device=dev, generator=gen)
  sets = [[mu_vec + s * torch.randn(d, device=dev, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()] for _ in range(2)]
  runners = {mode: AggregationStep(n, f, f, gar="krum", momentum=0.99, dampening=0.99, momentum_at="update", attack_evals=16,
                                   nb_past=25, line_search=mode) for mode in ("auto", "host")}
  times = {mode: [] for mode in runners}
  for i in range(27 + 40):
    for mode, runner in runners.items():
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      runner.run(sets[i & 1])
      runner.floats()
      dt = (time.perf_counter() - t0) * 1e3
      if i >= 27:
        times[mode].append(dt)
  assert runners["auto"].last_factor == runners["host"].last_factor and runners["auto"].floats() == runners["host"].floats()
  for mode, each in times.items():
    each = sorted(each)
    print(f"d={d} krum step with a 16-evaluation search, line_search={mode:5s}: median {each[len(each) // 2]:.4f} ms, mean "
          f"{sum(each) / len(each):.4f} ms, min {each[0]:.4f}, max {each[-1]:.4f} (factor {runners[mode].last_factor})", flush=True)


if __name__ == "__main__":
  main()
