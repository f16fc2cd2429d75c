"""The d-sized kernels of the factor searches added in ABI 22 / 23, five launches each at the bench's shapes (n = 25, f = 5,
d = 11 173 962, one tensor per row), for the FETCH_SIZE / WRITE_SIZE passes of rocprofv3 (scripts/gpu_call.sh searchpmc):
order_pair_kernel (h = 20), colwise_eval_kernel<3, MEDIAN>, sqdist2_kernel, bulyan_pass2_eval_kernel<25, 5, 4> under the ranking
of the candidate stack at factor 1.1 (the f copies of the candidate among the ranked rows read from cache: their table entries
point at avg).  Prints the algorithmic bytes of one launch of each (4-byte units of d)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import byzantinemomentum_amd as bm  # noqa: E402
from byzantinemomentum_amd import _lib, gars  # noqa: E402

dev = torch.device("cuda:0")
n, f, d = 25, 5, bench.D_RESNET18
h, m = n - f, n - f - 2
bench.SEPARATE_ROWS = True
honests = bench.make_stacks(n, f, d, dev, 1, 4321, False)[0][:h]
avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
cand = torch.empty_like(avg)
bm.stats.multi_fma3([cand], [avg], [direction], 1.0, 1.1)
order, _ = gars._rank(honests + [cand] * f, f, m, _lib.RANK_BULYAN)
copies = int((order[:m] >= h).sum())
for _ in range(5):
  lo, hi = bm.stats.order_pair(honests, (n - 1) // 2 - f, (n - 1) // 2)
  bm.stats.colwise_eval("median", [lo, hi], 1, 0, avg, direction, 1.1)
  bm.stats.sqdist2(lo, avg)
  bm.stats.bulyan_pass2_eval(honests, f, order, f, m, avg, direction, 1.1)
torch.cuda.synchronize()
unit = 4 * d
print(f"algorithmic bytes per launch: order_pair {unit * (h + 2)}  colwise_eval<3> {unit * 4}  sqdist2 {unit * 2}  "
      f"bulyan_pass2_eval {unit * (m - copies + 2)} ({m} ranked rows, {copies} of them copies of the candidate, + avg + dir)")
