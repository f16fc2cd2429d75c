"""Benchmark of the aggregation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload colwise|krum|bulyan|step [--gar RULE]]

Default workload = BASELINE.json configs[1]: coordinate-wise median + trimmed mean (f=5) over a
synthetic stack of n=25 worker gradients x d=11 173 962 coordinates (ResNet-18-sized), fp32, inputs
resident in HBM.  One "step" = one pass of the path over one batch = one median aggregation + one
trimmed-mean aggregation; `value` = aggregations per second over the whole job.

N > 1 (launched by torch.distributed.run, one rank per GPU): the path shards along d with no
data-path collective for the coordinate-wise rules — every rank aggregates its own d-slice of a
N-times-larger model ("weak" scaling, per-GPU work fixed); `value` counts one aggregation per
rank-shard pass.  `--workload bulyan` is the dim-sharded rule WITH its one real exchange (a single
all-reduce of the 25x25 fp64 squared-distance partials over RCCL).  `--workload step` is BASELINE.json
configs[4] on one GPU: the attack.py:800-878 mirror (worker momentum, empire attack, rule, study
statistics) at d = 36 546 980 (WRN-28-10 / CIFAR-100), n=25, f=5.

The JSON line also carries `roofline` (algorithmic bytes / HIP-event kernel time vs 8 TB/s HBM)
and `cpu_baseline` (the oracle's reference-faithful PyTorch-CPU port on this box's host cores,
rank 0, N=1 only).
"""

import argparse
import json
import os
import pathlib
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
D_RESNET18 = 11173962


def parse():
  p = argparse.ArgumentParser()
  p.add_argument("--gpus", type=int, default=1)
  p.add_argument("--steps", type=int, default=50)
  p.add_argument("--warmup", type=int, default=5)
  p.add_argument("--workload", default="colwise", choices=["colwise", "krum", "bulyan", "step"])
  p.add_argument("--gar", default="krum", help="aggregation rule of --workload step")
  p.add_argument("--d", type=int, default=D_RESNET18)
  p.add_argument("--no-cpu-baseline", action="store_true")
  p.add_argument("--aliased-byz", action="store_true",
                 help="make the f Byzantine rows ONE aliased tensor as the reference's attacks do "
                      "(attacks/identical.py:86); they are then served from cache and the HBM traffic "
                      "drops below the algorithmic bytes. Default: every row is a distinct buffer, so "
                      "that algorithmic bytes == bytes that must come from HBM.")
  return p.parse_args()


def make_stacks(n, f, d, device, count, seed, aliased):
  """`count` independent stacks (rotated between steps so that the 256 MB Infinity Cache never
  holds the next input). Honest rows N(mu, sigma_i); the f Byzantine rows are -0.1*mean(honest)
  ("empire", factor 1.1), either ONE aliased tensor (aliased=True, the reference's layout) or f
  distinct buffers with a 1e-3 relative jitter (default: every row costs its HBM bytes)."""
  gen = torch.Generator(device=device).manual_seed(seed)
  stacks = []
  for _ in range(count):
    mu = 0.1 * torch.randn(d, device=device, generator=gen)
    h = n - f
    sig = torch.linspace(0.5, 1.5, h).tolist()
    honest = [mu + s * torch.randn(d, device=device, generator=gen) for s in sig]
    byz = torch.stack(honest).mean(dim=0).mul_(-0.1)
    if aliased:
      stacks.append(honest + [byz] * f)
    else:
      stacks.append(honest + [byz + 1e-3 * byz.abs().mean() * torch.randn(d, device=device, generator=gen)
                              for _ in range(f)])
  return stacks


class KernelTimer:
  """HIP events (torch.cuda.Event on the stream the kernels are launched on) around each call."""

  def __init__(self):
    self.pairs = {}

  def run(self, name, fn):
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    out = fn()
    b.record()
    self.pairs.setdefault(name, []).append((a, b))
    return out

  def mean_ms(self, name):
    ps = self.pairs[name]
    return sum(a.elapsed_time(b) for a, b in ps) / len(ps)


def _pick_threads(fn):
  """The host has far more hardware threads than torch's CPU kernels can use on these shapes;
  time a small sample at a few thread counts and keep the fastest (reported as `cores`)."""
  total = os.cpu_count() or 1
  best, best_t = total, None
  for threads in sorted({min(total, c) for c in (16, 32, 64, total // 2, total)}):
    if threads < 1:
      continue
    torch.set_num_threads(threads)
    fn()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    if best_t is None or dt < best_t:
      best, best_t = threads, dt
  torch.set_num_threads(best)
  return best


def cpu_baseline_colwise(stack, f):
  from oracle import gar_oracle as O
  rows = [g.cpu() for g in stack]
  d = rows[0].shape[0]
  small = [r[:d // 16] for r in rows]
  threads = _pick_threads(lambda: (O.median(small), O.trmean(small, f)))
  t0 = time.perf_counter()
  reps = 2
  for _ in range(reps):
    O.median(rows)
    O.trmean(rows, f)
  dt = (time.perf_counter() - t0) / reps
  return {"value": 2.0 / dt, "unit": "agg/s", "cores": threads, "kind": "port",
          "sample": f"oracle f32 port (torch.stack+median, torch.stack+sort+mean: the reference's ops) on the same "
                    f"n={len(rows)} x d={d} stack, {reps} passes of median+trmean, {dt:.3f} s per pass, "
                    f"{threads} torch threads (fastest of 16/32/64/{(os.cpu_count() or 2) // 2}/{os.cpu_count()} "
                    f"on a d/16 sample; host has {os.cpu_count()} hardware threads)"}


def cpu_baseline_distance(stack, f, rule, d_sample):
  from oracle import gar_oracle as O
  n = len(stack)
  rows = [g[:d_sample].cpu() for g in stack]
  tiny = [r[:d_sample // 16] for r in rows]
  _pick_threads(lambda: O.krum(tiny, f))
  t0 = time.perf_counter()
  (O.krum if rule == "krum" else O.bulyan)(rows, f)
  dt = time.perf_counter() - t0
  scale = stack[0].shape[0] / d_sample
  return {"value": 1.0 / (dt * scale), "unit": "agg/s", "cores": torch.get_num_threads(), "kind": "port",
          "sample": f"oracle f32 port of {rule} on n={n} x d={d_sample} (first coordinates of the same stack), one "
                    f"pass {dt:.2f} s, scaled linearly to d={stack[0].shape[0]}"}


def main():
  args = parse()
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs an MI355X: no GPU visible")
  device = torch.device("cuda", local_rank)
  torch.cuda.set_device(device)
  # under torch.distributed.run the process group is created even for one rank, so that a single-GPU
  # run exercises exactly the RCCL code path of the N>1 runs
  distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
  if distributed:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=device)
  from byzantinemomentum_amd import build as bm_build
  if rank == 0:
    bm_build.build()  # no-op when the in-tree libbm_gar.so is current
  if distributed:
    dist.barrier()
  import byzantinemomentum_amd as bm
  bm._lib.load()

  d = args.d
  timer = KernelTimer()
  if args.workload == "colwise":
    n, f = 25, 5
    stacks = make_stacks(n, f, d, device, 2, 1234 + rank, args.aliased_byz)
    aggs_per_step = 2
    algo_bytes = {"median": 4 * d * (n + 1), "trmean": 4 * d * (n + 1)}

    def step(i, timed):
      st = stacks[i & 1]
      if timed:
        timer.run("median", lambda: bm.median(st))
        timer.run("trmean", lambda: bm.trmean(st, f))
      else:
        bm.median(st)
        bm.trmean(st, f)
    workload_name = f"C2 colwise: median + trmean(f={f}), n={n}, d={d} per GPU"
  elif args.workload == "step":
    from byzantinemomentum_amd.step import AggregationStep
    n, f = 25, 5
    h = n - f
    if args.d == D_RESNET18:
      d = 36546980  # WRN-28-10 on CIFAR-100
    runner = AggregationStep(n, f, f, gar=args.gar, momentum=0.99, dampening=0.99, attack_factor=1.1, nb_past=25)
    gen = torch.Generator(device=device).manual_seed(77 + rank)
    mu_vec = 0.1 * torch.randn(d, device=device, generator=gen)
    sets = [[mu_vec + s * torch.randn(d, device=device, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()]
            for _ in range(2)]
    aggs_per_step = 1
    m = n - f - 2
    gar_units = {"krum": n + m + 1, "bulyan": n + m + 1, "median": n + 1, "trmean": n + 1}.get(args.gar, n + 1)
    # 4-byte units of d per step: momentum 3h, attack+honest stats h+2, rule,
    # sampled stats h+1, attack stats 2, defense stats 1, dots 4+25
    algo_bytes = {"step": 4 * d * (3 * h + (h + 2) + gar_units + (h + 1) + 2 + 1 + 29)}

    def step(i, timed):
      if timed:
        timer.run("step", lambda: (runner.run(sets[i & 1]), runner.floats()))
      else:
        runner.run(sets[i & 1])
        runner.floats()
    workload_name = (f"C5 full step mirror (attack.py:800-878): worker momentum 0.99, empire 1.1, rule {args.gar}, "
                     f"study statistics with 25 past gradients; n={n}, f={f}, d={d}")
  else:
    from byzantinemomentum_amd.sharded import ShardedAggregator
    agg = ShardedAggregator(force_collectives=distributed)
    if args.workload == "krum":
      n, f = 51, 12
    else:
      n, f = 25, 5
    m = n - f - 2
    stacks = make_stacks(n, f, d, device, 2, 4321 + rank, args.aliased_byz)
    aggs_per_step = 1
    algo_bytes = {args.workload: 4 * d * n + 4 * d * (m + 1)}
    rule = agg.krum if args.workload == "krum" else agg.bulyan

    def step(i, timed):
      st = stacks[i & 1]
      if timed:
        timer.run(args.workload, lambda: rule(st, f))
      else:
        rule(st, f)
    workload_name = (f"{'C3 multi-krum' if args.workload == 'krum' else 'C4 bulyan'}: n={n}, f={f}, m={m}, "
                     f"d={d} per GPU, dim-sharded, one all-reduce of the {n}x{n} fp64 partial matrix")

  def barrier():
    if distributed:
      dist.barrier()
    torch.cuda.synchronize()

  for i in range(args.warmup):
    step(i, False)
  barrier()
  t0 = time.perf_counter()
  for i in range(args.steps):
    step(i, True)
  barrier()
  elapsed = time.perf_counter() - t0
  if distributed:
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()

  if rank == 0:
    per_kernel = {}
    for name, nbytes in algo_bytes.items():
      ms = timer.mean_ms(name)
      per_kernel[name] = {"avg_ms": ms, "algorithmic_bytes": nbytes, "gbps": nbytes / ms / 1e6,
                          "frac_of_8TBps": nbytes / ms / 1e6 / HBM_PEAK_GBPS, "agg_per_s": 1e3 / ms}
    dominant = max(per_kernel, key=lambda k: per_kernel[k]["avg_ms"])
    dk = per_kernel[dominant]
    traffic = None
    prof = ROOT / "profiles" / "pmc_traffic.json"
    if prof.exists():
      try:
        traffic = json.loads(prof.read_text()).get(args.workload, {}).get(dominant)
      except Exception:  # noqa: BLE001
        traffic = None
    line = {
      "metric": "aggregations/sec (Byzantine-robust GAR over n workers x d dims; achieved HBM GB/s per GAR in roofline/per_gar)",
      "value": aggs_per_step * args.steps * world / elapsed,
      "unit": "agg/s",
      "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": elapsed / args.steps * 1e3,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "f32", "data": "synthetic",
      "config": {"workload": workload_name, "n_workers": n, "f": f, "d_per_gpu": d,
                 "byzantine_rows": "aliased" if args.aliased_byz else "distinct buffers",
                 "parallelism": f"dim-shard x{world}" if world > 1 else "single GPU"},
      "roofline": {"bound": "hbm", "kernel": dominant, "achieved": dk["gbps"], "peak": HBM_PEAK_GBPS,
                   "unit": "GB/s", "frac": dk["frac_of_8TBps"], "traffic": traffic},
      "per_gar": per_kernel,
    }
    if world == 1 and not args.no_cpu_baseline:
      if args.workload == "colwise":
        line["cpu_baseline"] = cpu_baseline_colwise(stacks[0], f)
      elif args.workload == "step":
        pass  # the CPU baseline of the full step is the reference's attack.py itself
      else:
        line["cpu_baseline"] = cpu_baseline_distance(stacks[0], f, args.workload, min(d, 1 << 20))  # bounded sample
    print(json.dumps(line))
  if distributed:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
