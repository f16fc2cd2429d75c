"""One build of the library against another on the distance pass (A/B of a change to the Gram kernel):

    for i in 1 2 3; do python scripts/gram_variant_probe.py; BM_GAR_LIB=scratch/gram_parent/libbm_gar_gram_parent.so python scripts/gram_variant_probe.py; done

in ONE gpurun call (boxes differ by more than the effect).  The distance pass alone (Gram kernel + reduction + gated
launch, bm_pairwise_sqdist) at the C4 and C3 shapes, rows one torch.empty each, two alternating stacks."""
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import byzantinemomentum_amd as bm

dev = torch.device("cuda:0")
D = 11173962
tag = os.path.basename(os.environ.get("BM_GAR_LIB", "in-tree libbm_gar.so"))
bench.SEPARATE_ROWS = True
for n, f in ((25, 5), (51, 12)):
  stacks = bench.make_stacks(n, f, D, dev, 2, 4321, False)
  for i in range(4):
    sq = bm.gars.pairwise_sqdist(stacks[i & 1])
  torch.cuda.synchronize()
  us = []
  for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(20):
      sq = bm.gars.pairwise_sqdist(stacks[i & 1])
    b.record()
    torch.cuda.synchronize()
    us.append(a.elapsed_time(b) * 1e3 / 20)
  us.sort()
  print(f"{tag:34s} distance pass n={n}, d={D}: median of 5 rounds {us[2]:7.1f} us (best {us[0]:7.1f}); checksum {float(sq.sum()):.6f}", flush=True)
  del stacks
  torch.cuda.empty_cache()
