"""The same question for the first pass of a C5 step: skewing the sampled gradients and / or the momentum
buffers of AggregationStep by i x 256 B or i x 4 352 B per row against plain allocations.

    python scripts/step_skew_probe.py         # on the MI355X; profiles/r04_k_step_skew_probe.txt
"""
import os
import sys

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import byzantinemomentum_amd as bm
from byzantinemomentum_amd.step import AggregationStep
dev = torch.device('cuda:0')
n, f, d = 25, 5, 36546980
h = n - f
gen = torch.Generator(device=dev).manual_seed(77)
mu_vec = 0.1 * torch.randn(d, device=dev, generator=gen)

def skewed(count, d, step, zero=False):
    rows = []
    for i in range(count):
        buf = (torch.zeros if zero else torch.empty)(d + 64 * step + 1024, dtype=torch.float32, device=dev)
        rows.append(buf[i * step: i * step + d])
    return rows

def make_sampled(step):
    sets = []
    for _ in range(2):
        rows = skewed(h, d, step) if step else [torch.empty(d, device=dev) for _ in range(h)]
        for r, s in zip(rows, torch.linspace(0.5, 1.5, h).tolist()):
            r.copy_(mu_vec + s * torch.randn(d, device=dev, generator=gen))
        sets.append(rows)
    return sets

def run(gar, sampled_step, buffer_step):
    sets = make_sampled(sampled_step)
    runner = AggregationStep(n, f, f, gar=gar, momentum=0.99, dampening=0.99, attack_factor=1.1, nb_past=25)
    if buffer_step:
        runner._new_rows = staticmethod(lambda count, like, zero=False: skewed(count, like.shape[0], buffer_step, zero))
        runner._new_rows = lambda count, like, zero=False: skewed(count, like.shape[0], buffer_step, zero)
    for i in range(27):
        runner.run(sets[i & 1]); runner.floats()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(16):
        runner.run(sets[i & 1]); runner.floats()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 16

for rep in range(2):
    for gar in ("krum", "median"):
        for ss, bs in ((0, 0), (0, 1088), (1088, 0), (1088, 1088), (64, 64), (0, 0)):
            print(f"{gar:6s} sampled skew {ss*4:5d} B  buffer skew {bs*4:5d} B : {run(gar, ss, bs):.4f} ms/step", flush=True)
