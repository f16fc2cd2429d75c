"""What placement is: median / trimmed mean / Bulyan at n = 25, d = 11 173 962 on the same data with the rows
(a) one 2 MB-aligned torch.empty each, (b) the same with row i shifted by i x 256 B / 4 352 B / 17 KB inside its own
buffer, (c) cut out of one allocation at a skewed stride (layout.alloc_rows).  One process, alternating, two rounds.

    python scripts/row_skew_probe.py          # on the MI355X; profiles/r04_j_row_skew_probe.txt
"""
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import byzantinemomentum_amd as bm
dev = torch.device('cuda:0')
n, f, d = 25, 5, 11173962
bench.SEPARATE_ROWS = True
base = bench.make_stacks(n, f, d, dev, 2, 1234, False)
def timeit(fn, st):
    for i in range(5): fn(st[i & 1])
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(40): fn(st[i & 1])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 40 * 1e3
variants = {"separate": base}
for name, step in (("separate+skew 4352 B", 1088), ("separate+skew 17 KB", 4352), ("separate+skew 256 B", 64)):
    st = []
    for s in base:
        rows = []
        for i, g in enumerate(s):
            buf = torch.empty(d + 64 * step + 1024, dtype=torch.float32, device=dev)
            v = buf[i * step: i * step + d]
            v.copy_(g)
            rows.append(v)
        st.append(rows)
    variants[name] = st
bench.SEPARATE_ROWS = False
variants["slab (alloc_rows)"] = bench.make_stacks(n, f, d, dev, 2, 1234, False)
for rep in range(2):
    for name, st in variants.items():
        print(f"{name:24s} median {timeit(lambda s: bm.median(s), st):7.1f} us   trmean {timeit(lambda s: bm.trmean(s, f), st):7.1f} us   bulyan {timeit(lambda s: bm.bulyan(s, f), st):7.1f} us")
