"""bm_selected_mean: bit-exactness against torch's own sequential adds on the GPU and time per call, for the
plain and the burst form (BM_MEAN_BURST, read once per process: run once per value inside one gpurun call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm  # noqa: E402


def cases_from(argv, env):
  """Default cases (optionally only the first argv[1] of them), or BM_PROBE_CASES="n:m:d,n:m:d,..."."""
  if env.get("BM_PROBE_CASES"):
    return [tuple(int(x) for x in item.split(":")) for item in env["BM_PROBE_CASES"].split(",")]
  default = [(51, 37, 11173962), (25, 18, 36546980), (11, 7, 9000001), (25, 25, 11173962)]
  return default[:int(argv[1])] if len(argv) > 1 else default


def main():
  dev = torch.device("cuda:0")
  for n, m, d in cases_from(sys.argv, os.environ):
    gen = torch.Generator(device=dev).manual_seed(n)
    rows = [torch.randn(d, device=dev, generator=gen) for _ in range(n)]
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).tolist()
    idx = torch.tensor(perm + [0] * (64 - n), dtype=torch.int32, device=dev)
    out = bm.gars.selected_mean(rows, idx, m)
    acc = torch.zeros(d, device=dev)
    for i in perm[:m]:
      acc = acc + rows[i]
    want = acc / torch.tensor(float(m), device=dev)  # tensor / tensor: torch divides by a Python scalar through its reciprocal
    exact = bool(torch.equal(out, want))
    del acc, want
    torch.cuda.synchronize()
    reps, rounds = 20, 5
    us = []
    for _ in range(rounds):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for _ in range(reps):
        out = bm.gars.selected_mean(rows, idx, m)
      b.record()
      torch.cuda.synchronize()
      us.append(a.elapsed_time(b) * 1e3 / reps)
    us.sort()
    print(f"BM_MEAN_BURST={os.environ.get('BM_MEAN_BURST', 'default')} n={n} m={m} d={d}: bit-exact {exact}, sum {float(out.double().sum()):.9f}; "
          f"{us[rounds // 2]:.1f} us = {4 * d * (m + 1) / us[rounds // 2] / 1e3:.0f} GB/s")
    del rows
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()
