"""First-contact probe on the MI355X box: device facts, attainable HBM bandwidth, and the
column kernel (median / trmean) checked against torch-on-GPU and timed at config C2.

Usage: python scripts/gpu_probe.py [info|colwise] ...
Writes nothing itself; redirect stdout into gpurun_out/.
"""

import ctypes
import os
import pathlib
import subprocess
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
LIB = ROOT / "byzantinemomentum_amd" / "libbm_gar.so"


def ptr_table(ts):
  arr = (ctypes.c_void_p * len(ts))()
  for i, t in enumerate(ts):
    arr[i] = t.data_ptr()
  return arr


def timed(fn, iters=20, warm=3):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in evs:
    a.record()
    fn()
    b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in evs)
  return ts[len(ts) // 2], ts[0]


def info():
  p = torch.cuda.get_device_properties(0)
  print("device:", p.name, "CUs", p.multi_processor_count, "mem GiB", p.total_memory / 2**30,
        "gcn", getattr(p, "gcnArchName", "?"), "clock", getattr(p, "clock_rate", "?"))
  print("torch", torch.__version__, "hip", torch.version.hip)
  print("cpu_count", os.cpu_count())
  try:
    out = subprocess.run("lscpu | head -20", shell=True, capture_output=True, text=True).stdout
    print(out)
  except Exception as e:  # noqa
    print("lscpu failed", e)
  n = 2**29  # 2 GiB fp32
  a = torch.empty(n, device="cuda").normal_()
  b = torch.empty_like(a)
  med, best = timed(lambda: b.copy_(a), 10)
  print(f"copy 2GiB: median {med:.3f} ms -> {2 * 4 * n / med / 1e6:.1f} GB/s (best {2 * 4 * n / best / 1e6:.1f})")
  med, best = timed(lambda: a.sum(), 10)
  print(f"sum 2GiB: median {med:.3f} ms -> {4 * n / med / 1e6:.1f} GB/s (best {4 * n / best / 1e6:.1f})")
  med, best = timed(lambda: torch.add(a, b, out=b), 10)
  print(f"add 2GiB: median {med:.3f} ms -> {3 * 4 * n / med / 1e6:.1f} GB/s")


def colwise(n=25, d=11173962, f=5):
  lib = ctypes.CDLL(str(LIB))
  lib.bm_colwise.restype = ctypes.c_int
  lib.bm_colwise.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int64,
                             ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
  g = torch.Generator(device="cuda").manual_seed(1234)
  stacks = [[torch.randn(d, device="cuda", generator=g) for _ in range(n)] for _ in range(2)]
  out = torch.empty(d, device="cuda")
  stream = torch.cuda.current_stream().cuda_stream
  tabs = [ptr_table(s) for s in stacks]
  # correctness vs torch on the GPU
  for op, name in ((0, "median"), (1, "trmean"), (2, "phocas"), (3, "meamed")):
    rc = lib.bm_colwise(op, tabs[0], n, d, f, out.data_ptr(), stream)
    torch.cuda.synchronize()
    st = torch.stack(stacks[0])
    if name == "median":
      ref = st.median(dim=0)[0]
    elif name == "trmean":
      ref = st.sort(dim=0).values[f:-f].mean(dim=0)
    else:
      c = st.median(dim=0)[0] if name == "meamed" else st.sort(dim=0).values[f:-f].mean(dim=0)
      m = n - f
      p = st.clone().sub_(c).abs_().topk(m, dim=0, largest=False, sorted=False).indices
      p.mul_(d).add_(torch.arange(0, d, dtype=p.dtype, device=p.device))
      ref = st.take(p).mean(dim=0)
    err = (out - ref).abs().max().item()
    print(f"{name}: rc={rc} max|diff| vs torch-gpu = {err:.3e} equal={torch.equal(out, ref)}")
    del st, ref
  # timing, alternating two stacks (each 1.1 GB > 256 MB Infinity Cache)
  algo = 4 * d * (n + 1)
  for op, name in ((0, "median"), (1, "trmean"), (2, "phocas"), (3, "meamed")):
    k = [0]

    def run():
      lib.bm_colwise(op, tabs[k[0] & 1], n, d, f, out.data_ptr(), stream)
      k[0] += 1

    med, best = timed(run, 30)
    print(f"{name} n={n} d={d}: median {med * 1e3:.1f} us  {algo / med / 1e6:.1f} GB/s "
          f"({algo / med / 1e6 / 8000 * 100:.1f}% of 8 TB/s)  best {algo / best / 1e6:.1f} GB/s  "
          f"{1e3 / med:.0f} agg/s   [BM_FORCE_VEC={os.environ.get('BM_FORCE_VEC', '')} "
          f"BM_COL_MAX_BLOCKS={os.environ.get('BM_COL_MAX_BLOCKS', '')}]")
  # torch-on-GPU reference timing (second baseline)
  st = None
  t0 = time.time()
  med, best = timed(lambda: torch.stack(stacks[0]).median(dim=0)[0], 3, 1)
  print(f"torch-gpu stack+median: {med:.2f} ms")
  med, best = timed(lambda: torch.stack(stacks[0]).sort(dim=0).values[f:-f].mean(dim=0), 3, 1)
  print(f"torch-gpu stack+sort+mean (trmean): {med:.2f} ms")


if __name__ == "__main__":
  what = sys.argv[1] if len(sys.argv) > 1 else "info"
  if what == "info":
    info()
  elif what == "colwise":
    args = [int(a) for a in sys.argv[2:]]
    colwise(*args)
