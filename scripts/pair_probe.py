"""Time bm_pairwise_sqdist alone (HIP events), for experiments with the BM_PAIR_* knobs."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm

def main():
  d = 11173962
  for n in (25, 51):
    gen = torch.Generator(device="cuda").manual_seed(n)
    stacks = [[torch.randn(d, device="cuda", generator=gen) for _ in range(n)] for _ in range(2)]
    for i in range(3):
      bm.gars.pairwise_sqdist(stacks[i & 1])
    torch.cuda.synchronize()
    evs = []
    for i in range(10):
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record(); bm.gars.pairwise_sqdist(stacks[i & 1]); b.record()
      evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    sq = bm.gars.pairwise_sqdist(stacks[0]).cpu()
    errs = []
    for (i, j) in ((0, 1), (2, n - 1), (n // 2, n // 2 + 3)):
      want = (stacks[0][i].double() - stacks[0][j].double()).pow(2).sum().item()
      errs.append(abs(sq[i, j].item() - want) / want)
    algo = 4 * d * n
    print(f"n={n} pairwise median {ts[5]*1e3:.0f} us best {ts[0]*1e3:.0f} us  {algo/ts[5]/1e6:.0f} GB/s "
          f"maxrelerr {max(errs):.1e} [{' '.join(k + '=' + v for k, v in os.environ.items() if k.startswith('BM_'))}]", flush=True)

main()
