"""Benchmark of the aggregation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload colwise|krum|bulyan|step [--gar RULE]]

N = 1 (default): BASELINE.json configs[1] — coordinate-wise median + trimmed mean (f=5) over a synthetic
stack of n=25 worker gradients x d=11 173 962 coordinates (ResNet-18-sized), fp32, inputs resident in
HBM.  One "step" = one pass of the path over one batch = one median + one trimmed-mean aggregation;
`value` = aggregations per second.  The same line carries, under `per_gar`, short measurements of the
other single-GPU configurations (C3 Multi-Krum n=51, C4 Bulyan n=25, C5 full step at d=36.5 M), each
with its algorithmic bytes and fraction of the 8 TB/s HBM roofline (skip with --no-extras).

N > 1 (one rank per GPU; `python bench.py --gpus N` re-executes itself under torch.distributed.run, and the driver's own
torchrun launch is used as is): the SAME workload, the gradient dimension partitioned across the ranks: every rank
holds a shard of 11 173 962 coordinates of the n = 25 gradients of an N x 11 173 962-coordinate vector ("weak" scaling:
the work per GPU is that of the one-GPU line).  The coordinate-wise rules are independent per coordinate, so this
path has no exchange step and no collective is issued inside the timed region; `value` = aggregations of an
n = 25 x d = 11 173 962 stack per second over all ranks (the unit of the one-GPU line).  The path's one real exchange
is measured in the same run, under `per_gar`: BASELINE.json configs[3] — Bulyan n=25, f=5 over a FIXED total
d = 11 173 962 split across the ranks by shard_bounds ("strong" scaling), with the all-reduce of the 25x25 fp64
squared-distance partials over RCCL inside the timed call (`bulyan_c4_sharded`, next to `single_gpu_same_workload`,
the unsharded rule timed on rank 0 alone), the optional all-gather of the output, and the worker-parallel layout
exchange.  `--workload bulyan|krum|step` makes the sharded rule / step the timed workload itself (strong scaling
unless --scaling weak).

The JSON line also carries `roofline` (algorithmic bytes / HIP-event time of the dominant kernel vs
8 TB/s; `traffic` = HBM bytes per launch from rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this very
command, collected live unless --no-traffic) and `cpu_baseline` (the reference's own
aggregators — the staged checkout of scripts/stage_reference.sh — or, without one, the oracle's pinned f32 port, on this
box's host cores, rank 0, N = 1 only; full-size stacks, nothing extrapolated).
"""

import argparse
import csv
import glob
import json
import os
import pathlib
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
D_RESNET18 = 11173962
D_WRN = 36546980         # WRN-28-10 on CIFAR-100


def parse():
  p = argparse.ArgumentParser()
  p.add_argument("--gpus", type=int, default=1)
  p.add_argument("--steps", type=int, default=50)
  p.add_argument("--warmup", type=int, default=5)
  p.add_argument("--workload", default="colwise", choices=["colwise", "krum", "bulyan", "step"],
                 help="default: colwise (BASELINE configs[1]) at every N; krum / bulyan / step: the dim-sharded rule or step")
  p.add_argument("--scaling", default=None, choices=["weak", "strong"],
                 help="N > 1: weak = every rank holds --d coordinates (default for colwise, which has no exchange step), "
                      "strong = --d coordinates in total, split by shard_bounds (default for krum / bulyan / step)")
  p.add_argument("--gar", default="krum", help="aggregation rule of --workload step")
  # (--dim: the spelling that survives `python -m torch.distributed.run ... bench.py --dim N`, whose own parser takes a
  #  bare --d for an abbreviation of its --duplicate-* options)
  p.add_argument("--d", "--dim", dest="d", type=int, default=None, help="TOTAL number of coordinates (split across the ranks)")
  p.add_argument("--weak", action="store_true", help="same as --scaling weak")
  p.add_argument("--no-cpu-baseline", action="store_true")
  p.add_argument("--no-extras", action="store_true", help="N = 1: skip the per_gar measurements of C3/C4/C5")
  p.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 FETCH_SIZE/WRITE_SIZE passes")
  p.add_argument("--slab-rows", action="store_true",
                 help="cut the rows of every synthetic stack out of ONE allocation at a skewed stride (byzantinemomentum_amd."
                      "layout.alloc_rows) instead of one torch.empty per gradient.  The default is what a drop-in caller has "
                      "(attack.py:676,803-804: one tensor per worker, wherever its allocator put it); the slab measured "
                      "+3 %% on the column kernels in round 3 and is reported as a side entry (per_gar.*_slab_rows).")
  p.add_argument("--separate-rows", action="store_true", help="(the default since round 4; kept for old command lines)")
  p.add_argument("--sharded-extras", action="store_true",
                 help="one rank under torch.distributed.run: run the per_gar legs of the N > 1 line (sharded Bulyan with "
                      "its all-reduce, all-gather, layout exchange) as well, so that their code is exercised on one GPU")
  p.add_argument("--extras-timeout", type=float, default=180.0,
                 help="N > 1: seconds the exchange legs that follow the timed headline (the library's own RCCL communicator, the "
                      "sharded rule, all-gather, layout exchange) may take on every rank before the line is printed without them "
                      "(`exchange.error`) and the job ends: a hang there must not cost the headline (0 = no limit)")
  p.add_argument("--graph-replay", action="store_true",
                 help="N > 1 (or one rank under torch.distributed.run): also time the sharded rule recorded into a HIP "
                      "graph, one graph per synthetic stack (byzantinemomentum_amd/graphs.py) -> per_gar.<rule>_graph_replay")
  p.add_argument("--aliased-byz", action="store_true",
                 help="make the f Byzantine rows ONE aliased tensor as the reference's attacks do "
                      "(attacks/identical.py:86); they are then served from cache and the HBM traffic "
                      "drops below the algorithmic bytes. Default: every row is a distinct buffer, so "
                      "that algorithmic bytes == bytes that must come from HBM.")
  return p.parse_args()


def make_stacks(n, f, d, device, count, seed, aliased):
  """`count` independent stacks (rotated between steps so that the 256 MB Infinity Cache never
  holds the next input). Honest rows N(mu, sigma_i); the f Byzantine rows are -0.1*mean(honest)
  ("empire", factor 1.1), either ONE aliased tensor (aliased=True, the reference's layout: exact zero
  distances between them) or f distinct buffers = that vector plus independent N(0, 0.3^2) noise
  (default: every row costs its HBM bytes and no two rows are near-duplicates — near-duplicate but
  not identical rows are handed to the exact direct-difference kernel by the accuracy gate of the
  distance pass, which is the slow path and not what this benchmark is meant to time)."""
  gen = torch.Generator(device=device).manual_seed(seed)
  stacks = []
  for _ in range(count):
    mu = 0.1 * torch.randn(d, device=device, generator=gen)
    h = n - f
    sig = torch.linspace(0.5, 1.5, h).tolist()
    rows = new_rows(n if not aliased else h + 1, d, device)
    for i, s in enumerate(sig):
      rows[i].copy_(mu + s * torch.randn(d, device=device, generator=gen))
    honest = rows[:h]
    byz = torch.stack(honest).mean(dim=0).mul_(-0.1)
    if aliased:
      rows[h].copy_(byz)
      stacks.append(honest + [rows[h]] * f)
    else:
      for j in range(f):
        rows[h + j].copy_(byz + 0.3 * torch.randn(d, device=device, generator=gen))
      stacks.append(list(rows))
  return stacks


SEPARATE_ROWS = True  # one torch.empty per row (the drop-in caller's layout); --slab-rows: layout.alloc_rows


def new_rows(count, d, device):
  """Row buffers of a synthetic stack: placed by the package's allocator (rows of one allocation at a skewed stride,
  byzantinemomentum_amd/layout.py) unless --separate-rows asks for one allocation per row."""
  if SEPARATE_ROWS:
    return [torch.empty(d, dtype=torch.float32, device=device) for _ in range(count)]
  from byzantinemomentum_amd.layout import alloc_rows
  return alloc_rows(count, d, device)


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
    """Mean over the calls — a float that also carries their median (`.median`): on a loaded host one call in ten or
    twenty can be held up for 10-130 ms between two launches (seen in round 6: profiles/r06_search_each.txt), which a
    mean over a dozen sub-millisecond calls turns into a different kernel."""
    ps = self.pairs[name]
    each = sorted(a.elapsed_time(b) for a, b in ps)
    out = Milliseconds(sum(each) / len(each))
    out.median = each[len(each) // 2]
    out.slowest = each[-1]
    out.held_up = sum(1 for v in each if v > 3.0 * out.median)  # calls that took more than three times the median call
    out.calls = len(each)
    return out


class Milliseconds(float):
  median = None
  slowest = None
  held_up = None
  calls = None


def entry(ms, nbytes, gpus=1, units=1, **more):
  """One per_gar record: `nbytes` = algorithmic bytes moved by ALL `gpus` ranks in `ms`; `units` = aggregations (of the
  metric's n x d stack) that completes; the roofline fraction is against the HBM peak of the GPUs involved."""
  rec = {"avg_ms": float(ms), "algorithmic_bytes": nbytes, "gbps": nbytes / ms / 1e6,
         "frac_of_8TBps": nbytes / ms / 1e6 / (HBM_PEAK_GBPS * gpus), "agg_per_s": units * 1e3 / ms}
  median = getattr(ms, "median", None)
  if median:  # the same figures on the median call (robust against a host that stalls between two launches)
    rec.update(median_ms=median, frac_of_8TBps_median=nbytes / median / 1e6 / (HBM_PEAK_GBPS * gpus))
    if getattr(ms, "held_up", None):  # why `avg_ms` and `median_ms` disagree: a call or two held up on the host
      rec.update(held_up_calls=ms.held_up, timed_calls=ms.calls, slowest_ms=ms.slowest)
  return dict(rec, **more)


def timed_loop(fn, steps, warmup, timer, name):
  """`steps` timed calls of fn(i) behind `warmup` untimed ones.  The ranking cache of the host mirror (gars.py: `influence`
  right after `aggregate` reuses the ranking, attack.py:821-822) is emptied before EVERY call: with an odd warm-up the
  first timed call of a two-stack rotation met the stack of the last warm-up call, skipped its distance pass and pulled
  the mean of a dozen calls 4 % down (Krum, Bulyan, Brute, Aksel entries of rounds 2-5; found in round 6 when the
  per-call medians came out ABOVE the means)."""
  from byzantinemomentum_amd import gars
  for i in range(warmup):
    gars.invalidate_rank_cache()
    fn(i)
  torch.cuda.synchronize()
  for i in range(steps):
    gars.invalidate_rank_cache()
    timer.run(name, lambda: fn(i))
  torch.cuda.synchronize()
  return timer.mean_ms(name)


# ---------------------------------------------------------------------------- #
# CPU baseline (rank 0, one GPU only): the oracle's f32 port = the reference's torch-CPU operations

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


def _cpu_reference():
  """The reference's own registry (`aggregators.gars`) when a checkout is importable — /root/reference in the build
  container, the staged oracle/_ref/reference (scripts/stage_reference.sh) on the GPU box — else None: the oracle's
  f32 port, pinned bit-identical to it by tests/test_oracle_vs_reference.py, is timed instead (kind "port")."""
  from oracle import reference_loader
  if not reference_loader.available():
    return None
  try:
    return reference_loader.load(with_native=False)[0].gars
  except Exception as err:  # noqa: BLE001  (a baseline must not take the line down)
    print(f"[bench] reference checkout present but not importable: {err!r}", file=sys.stderr)
    return None


def _host_facts():
  """CPU model, hardware threads and torch version of the box the baseline runs on (BASELINE.md section 2 asks for them)."""
  model = None
  try:
    with open("/proc/cpuinfo") as fh:
      for line in fh:
        if line.lower().startswith("model name"):
          model = line.split(":", 1)[1].strip()
          break
  except OSError:
    pass
  return {"cpu_model": model, "hardware_threads": os.cpu_count(), "torch": torch.__version__}


def _host_copy(stack):
  seen = {}
  return [seen.setdefault(id(g), g.cpu()) for g in stack]


def cpu_baseline_colwise(stack, f):
  """median.py:31-39 + trmean.py:69-79 on the SAME full-size stack, on this box's host cores."""
  from oracle import gar_oracle as O
  gars = _cpu_reference()
  rows = _host_copy(stack)
  d = rows[0].shape[0]
  if gars is not None:
    def pair(rs):
      gars["median"].unchecked(gradients=rs, f=f)
      gars["trmean"].unchecked(gradients=rs, f=f)
    what = "the reference's aggregators/median.py + trmean.py (gar.unchecked)"
  else:
    def pair(rs):
      O.median(rs)
      O.trmean(rs, f)
    what = "oracle f32 port (torch.stack+median, torch.stack+sort+mean: the reference's ops)"
  small = [r[:d // 16] for r in rows]
  threads = _pick_threads(lambda: pair(small))
  t0 = time.perf_counter()
  reps = 2
  for _ in range(reps):
    pair(rows)
  dt = (time.perf_counter() - t0) / reps
  return {"value": 2.0 / dt, "unit": "agg/s", "cores": threads, "kind": "reference" if gars is not None else "port", "host": _host_facts(),
          "sample": f"{what} on the same n={len(rows)} x d={d} stack (full size, nothing scaled), {reps} passes of "
                    f"median+trmean, {dt:.3f} s per pass, {threads} torch threads (fastest of 16/32/64/"
                    f"{(os.cpu_count() or 2) // 2}/{os.cpu_count()} on a d/16 sample; host has {os.cpu_count()} "
                    f"hardware threads)"}


def cpu_baseline_rule(rows, f, rule):
  """krum.py:31-80 / bulyan.py:31-84 on the host copy `rows` of the SAME full-size stack: ONE aggregation, nothing
  extrapolated.  A fixed thread count: the per-pair torch ops (`sub().norm().item()`) do not scale past it, and
  searching the count on them once picked one that made the run take minutes."""
  from oracle import gar_oracle as O
  gars = _cpu_reference()
  threads = min(32, os.cpu_count() or 1)
  torch.set_num_threads(threads)
  n, d = len(rows), rows[0].shape[0]
  if gars is not None:
    fn = lambda rs: gars[rule].unchecked(gradients=rs, f=f)  # noqa: E731
    what = f"the reference's aggregators/{rule}.py (gar.unchecked)"
  else:
    fn = lambda rs: (O.krum if rule == "krum" else O.bulyan)(rs, f)  # noqa: E731
    what = f"oracle f32 port of {rule}"
  fn([r[:4096] for r in rows])  # (first-use costs of the torch ops and of the thread pool stay out of the timing)
  t0 = time.perf_counter()
  fn(rows)
  dt = time.perf_counter() - t0
  return {"value": 1.0 / dt, "unit": "agg/s", "cores": threads, "kind": "reference" if gars is not None else "port", "host": _host_facts(),
          "sample": f"{what} on the same n={n}, f={f}, d={d} stack (full size, nothing scaled), one aggregation "
                    f"{dt:.2f} s on {threads} torch threads"}


def cpu_baseline_step(sampled, n, f, gar):
  """ONE step of the loop body attack.py:786-868 at full size on the host: oracle/step_oracle.ReferenceLoop in its
  f32 form (the reference's own fp32 torch-CPU operations in the reference's order; pinned bit-faithful against the
  loop body driven with the reference's own functions by tests/test_step_reference_vs_reference.py)."""
  from oracle.step_oracle import ReferenceLoop
  threads = min(32, os.cpu_count() or 1)
  torch.set_num_threads(threads)
  rows = [g.cpu() for g in sampled]
  d = rows[0].shape[0]
  loop = ReferenceLoop(n, f, f, gar, "worker", 0.99, 0.99, "empire", 1.1, None, 25, precision="f32")
  params, origin = torch.zeros(d), torch.zeros(d)
  t0 = time.perf_counter()
  loop.step(rows, params, origin)
  dt = time.perf_counter() - t0
  return {"value": 1.0 / dt, "unit": "steps/s", "cores": threads, "kind": "port", "host": _host_facts(),
          "sample": f"one full-size step (n={n}, f={f}, d={d}, worker momentum 0.99, empire 1.1, rule {gar}, study "
                    f"block; the first step of a run: no past gradient in the deque yet) of the f32 loop-body "
                    f"restatement, {dt:.2f} s on {threads} torch threads"}


# ---------------------------------------------------------------------------- #
# Live HBM traffic: rocprofv3 PMC passes of this very command (separate runs, --kernel-trace only)

def measure_traffic(argv, kernel_substrings, extras=False):
  """HBM bytes per launch of every kernel whose name contains one of `kernel_substrings` (dict key -> substring, or
  (substring, lowest grid size, highest grid size) in work-items when one kernel runs at two lengths in the command):
  FETCH_SIZE and WRITE_SIZE (KiB) from two rocprofv3 --pmc passes (their own runs, --kernel-trace only) of this very
  command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950.
  Returns {key: bytes} (keys whose kernel was not launched are absent); None if rocprofv3 is unavailable."""
  exe = shutil.which("rocprofv3")
  if exe is None:
    return None
  values = {}
  env = dict(os.environ, TMPDIR="/tmp", BM_BENCH_CHILD="1")
  for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
      cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--",
             sys.executable, str(ROOT / "bench.py"), *argv, "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
             "--no-traffic"] + ([] if extras else ["--no-extras"])
      try:
        subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, timeout=420, check=True)
      except Exception:  # noqa: BLE001
        return None
      sums, counts = {}, {}
      for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
          per_dispatch = {}
          for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
              continue
            for key, sub in kernel_substrings.items():
              if isinstance(sub, tuple):
                sub, lo, hi = sub
                try:
                  if not lo <= int(row["Grid_Size"]) <= hi:
                    continue
                except (KeyError, ValueError):
                  pass
              if sub in row["Kernel_Name"]:
                slot = (key, row["Dispatch_Id"])
                per_dispatch[slot] = per_dispatch.get(slot, 0.0) + float(row["Counter_Value"])
          for (key, _), val in per_dispatch.items():
            sums[key] = sums.get(key, 0.0) + val
            counts[key] = counts.get(key, 0) + 1
      values[counter] = {key: sums[key] / counts[key] for key in sums}
  return {key: int(1024 * (2 * values["FETCH_SIZE"][key] + values["WRITE_SIZE"][key]))
          for key in values["FETCH_SIZE"] if key in values["WRITE_SIZE"]}


# kernel-name substrings of the dominant kernel(s) behind each per_gar entry, with the algorithmic bytes of ONE launch
def traffic_kernels(d2, d5):
  return {
    "median": ("colwise_burst_kernel<25, 0", 4 * d2 * 26),
    "trmean": ("colwise_burst_kernel<25, 1", 4 * d2 * 26),
    "krum_c3.distances": ("gram3_partial_kernel<13, 2", 4 * d2 * 51),
    # (bulyan_pass2_kernel<25, 5, 4> runs at both lengths in the child command: told apart by the grid, 256 lanes per
    #  workgroup, one workgroup per 256 column groups up to 16 384 workgroups)
    "bulyan_c4_1gpu.pass2": (("bulyan_pass2_kernel<25, 5, 4>", 0, 256 * 16383), 4 * d2 * 19),
    "step_c5.first_pass": ("momentum_gram_kernel", 4 * d5 * 63),  # (Krum / Bulyan step: first pass + distance pass in one kernel)
    # the second pass of the Bulyan step: 18 ranked rows of which the f = 5 Byzantine ones are ONE buffer (the empire
    # vector ranks first): 14 distinct rows read + 1 written — what HBM delivers — against 19 counted row by row
    "step_c5.bulyan_pass2": (("bulyan_pass2_kernel<25, 5, 4>", 256 * 16384, 1 << 40), 4 * d5 * 15),
    "step_c5.study": ("study_stats_burst_kernel<true, 3, false, false>", 4 * d5 * 8),
  }


# ---------------------------------------------------------------------------- #

def sharded_extras(bm, agg, dist, device, world, rank, timer, args, rule_name, d_total, extra, time_rule, out):
  """The dim-sharded path with its one real exchange, on a FIXED vector of d_total coordinates split across the ranks by
  shard_bounds (strong scaling): the rule in one C call with the all-reduce of the n x n fp64 partials inside
  (BASELINE.json configs[3] for Bulyan), the optional all-gather of the output, the worker-parallel layout exchange, and
  the unsharded rule on rank 0 alone for the speed-up of this exact workload.  Every figure is the max over the ranks.
  The per_gar entries go into `out` and the top-level `exchange` object into `extra` leg by leg, so that a deadline that
  fires in a later leg keeps what the earlier ones measured; `extra` also receives single_gpu_same_workload."""
  from byzantinemomentum_amd.sharded import Shards, owned_workers, shard_bounds
  n, f = (51, 12) if rule_name == "krum" else (25, 5)
  m = n - f - 2
  # the path's one real exchange, at the top level of the line (a SCALE record then carries it, not only the
  # embarrassingly parallel headline): BASELINE.json configs[3] for Bulyan — strong scaling of a fixed d
  exchange = extra.setdefault("exchange", {})  # (the same object all along: a deadline may be writing its `error` into it)
  exchange.update({
    "workload": f"{rule_name} n={n} f={f}, total d={d_total} dim-sharded over {world} ranks (strong scaling), one all-reduce "
                f"of the {n}x{n} fp64 squared-distance partials inside the call",
    "ms": None, "agg_per_s": None, "allreduce_us": None, "allreduce_bytes": 8 * n * n, "allgather_output_ms": None,
    "layout_exchange_ms": None, "single_gpu_ms": None, "speedup_vs_1gpu": None,
    "collectives": "libbm_gar's own RCCL communicator" if agg.native is not None else "torch.distributed (RCCL)"})
  lo, hi = shard_bounds(d_total, world, rank)
  # (Shards: the shards state the length of the whole vectors, whatever other lengths this aggregator has served)
  stacks = [Shards(st, d_total=d_total) for st in make_stacks(n, f, hi - lo, device, 2, 4321 + rank, args.aliased_byz)]
  rule = agg.krum if rule_name == "krum" else agg.bulyan
  rule_bytes = 4 * d_total * n + 4 * d_total * (m + 1)
  tag = f"n={n}, f={f}, total d={d_total} over {world} ranks, max over ranks"

  def over_ranks(ms):
    t = torch.tensor([ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()

  if time_rule:
    # like the headline: W warm-up calls, then K calls between two barriers, wall clock, max over the ranks
    for i in range(max(args.warmup, 3)):
      rule(stacks[i & 1], f)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
      rule(stacks[i & 1], f)
    dist.barrier()
    torch.cuda.synchronize()
    ms = over_ranks((time.perf_counter() - t0) / args.steps * 1e3)
    out[f"{rule_name}_{'c3' if rule_name == 'krum' else 'c4'}_sharded"] = entry(
      ms, rule_bytes, gpus=world, config=f"{rule_name}, one C call per aggregation with the all-reduce of the {n}x{n} fp64 partial "
                                         f"matrix inside ({'libbm_gar RCCL communicator' if agg.native is not None else 'torch.distributed'}), "
                                         f"{tag}; wall clock of {args.steps} calls between barriers", scaling="strong")
    exchange["ms"], exchange["agg_per_s"] = ms, 1e3 / ms
  result = rule(stacks[0], f)
  # the exchange itself: the n x n fp64 matrix through the same collective the rule uses (latency-bound: 5 KB)
  probe = torch.zeros((n, n), dtype=torch.float64, device=device)
  if agg.native is not None:
    lib = bm._lib.load()
    reduce_once = lambda i: bm._lib.check(lib.bm_allreduce_sum_f64(  # noqa: E731
      agg.native.handle, probe.data_ptr(), n * n, torch.cuda.current_stream().cuda_stream), "bm_allreduce_sum_f64")
  else:
    reduce_once = lambda i: agg.all_reduce_sum(probe)  # noqa: E731
  us_ar = over_ranks(timed_loop(reduce_once, 50, 5, timer, "x_allreduce")) * 1e3
  exchange["allreduce_us"] = us_ar
  ms3 = over_ranks(timed_loop(lambda i: agg.all_gather_output(result, d_total), 10, 2, timer, "x_allgather"))
  exchange["allgather_output_ms"] = ms3
  out["allgather_output"] = entry(ms3, 4 * d_total, gpus=world, config=f"all-gather of the output slices, {tag}")
  if args.graph_replay:
    # the launch-bound regime: the whole aggregation — kernels and the all-reduce — as ONE graph launch.
    # Every rank votes after its recording, so that no rank replays (and enters the collective) alone.
    from byzantinemomentum_amd.graphs import GraphedCall
    graphs, failure = None, None
    try:
      graphs = [GraphedCall(lambda s=s: rule(stacks[s], f)) for s in (0, 1)]
    except Exception as err:  # noqa: BLE001
      failure = repr(err)
    vote = torch.tensor([0.0 if graphs is None else 1.0], dtype=torch.float64, device=device)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)
    if vote.item() >= 1.0:
      same = bool(torch.equal(graphs[0](), rule(stacks[0], f)))
      ms_g = over_ranks(timed_loop(lambda i: graphs[i & 1](), 20, 3, timer, "x_graph"))
      out[rule_name + "_graph_replay"] = dict(entry(ms_g, rule_bytes, gpus=world,
                                                   config=f"the same sharded rule replayed from a HIP graph, {tag}"),
                                              same_bits_as_eager=same)
    else:
      out[rule_name + "_graph_replay"] = {"error": failure or "the recording failed on another rank"}
    del graphs
  del stacks, result
  torch.cuda.empty_cache()
  single = None
  if rank == 0:
    full = make_stacks(n, f, d_total, device, 2, 4321, args.aliased_byz)
    fn = bm.krum if rule_name == "krum" else bm.bulyan
    single = timed_loop(lambda i: fn(full[i & 1], f), 10, 3, timer, "x_single")
    del full
    torch.cuda.empty_cache()
  dist.barrier()
  if rank == 0:
    extra["single_gpu_same_workload"] = {"value": 1e3 / single, "unit": "agg/s", "ms": single,
                                         "note": f"the unsharded {rule_name} on rank 0 alone, same total d = {d_total}"}
    exchange["single_gpu_ms"] = single
    if exchange["ms"] is not None:
      exchange["speedup_vs_1gpu"] = single / exchange["ms"]
  # worker-parallel production: every rank holds the FULL-length gradients of its own workers
  # (rank p runs workers p, p + P, ...); ONE all-to-all turns worker-major into dimension-major, then the rule
  mine = owned_workers(n, world, rank)
  gen = torch.Generator(device=device).manual_seed(999 + rank)
  produced = new_rows(max(len(mine), 1), d_total, device)[:len(mine)]
  for r in produced:
    r.copy_(0.1 * torch.randn(d_total, device=device, generator=gen))
  ms_a2a = over_ranks(timed_loop(lambda i: agg.to_dim_sharded(produced, n, d_total), 8, 2, timer, "x_a2a"))
  exchange["layout_exchange_ms"] = ms_a2a
  ms_wp = over_ranks(timed_loop(lambda i: rule(agg.to_dim_sharded(produced, n, d_total), f), 8, 2, timer, "x_wp"))
  sent = 4 * d_total * len(owned_workers(n, world, 0)) * (world - 1) // world
  for key, val in (("layout_exchange", ms_a2a), (rule_name + "_from_worker_parallel", ms_wp)):
    out[key] = dict(entry(val, 4 * d_total * n + (4 * d_total * (m + 1) if key != "layout_exchange" else 0), gpus=world,
                          config=f"worker-major -> dimension-major by one all-to-all (RCCL), {tag}"
                                 + ("" if key == "layout_exchange" else f", then {rule_name}")),
                    bytes_sent_per_rank=sent)
  del produced
  torch.cuda.empty_cache()
  return out


def make_aggregator(bm, dist, device, world, rank, distributed):
  """The sharded aggregator of this job.  With more than one rank (or one rank under torch.distributed.run) the HIP
  backend gets the library's own RCCL communicator, which must work on EVERY rank or be left by every rank together (a
  rank falling back alone would issue different collectives): one tiny all-reduce through it, then a vote."""
  from byzantinemomentum_amd.sharded import ShardedAggregator
  agg = ShardedAggregator(force_collectives=distributed)
  if distributed and agg.native is not None:
    ok = 1.0
    try:
      probe = torch.ones(4, dtype=torch.float64, device=device)
      bm._lib.check(bm._lib.load().bm_allreduce_sum_f64(agg.native.handle, probe.data_ptr(), 4,
                                                         torch.cuda.current_stream().cuda_stream), "bm_allreduce_sum_f64")
      torch.cuda.synchronize()
      ok = 1.0 if float(probe[0].item()) == float(world) else 0.0
    except Exception as err:  # noqa: BLE001
      print(f"[bench rank {rank}] native communicator failed its probe: {err}", file=sys.stderr)
      ok = 0.0
    vote = torch.tensor([ok], dtype=torch.float64, device=device)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)
    if vote.item() < 1.0:
      agg = ShardedAggregator(force_collectives=distributed, native_comm=False)
  return agg


class Deadline:
  """A limit on a leg that must not cost the line: when `seconds` pass before `cancel()`, `emit()` runs (rank 0
  prints what it has) and the process ends at once with status 0 — on every rank at about the same moment (they arm
  their deadlines right after a barrier), so that no rank waits in a collective for peers that left.  A thread, not a
  signal: a rank stuck inside RCCL or a ctypes call runs no Python signal handler, but both release the GIL."""

  def __init__(self, seconds, emit):
    import threading
    self.lock = threading.Lock()
    self.done = False
    self.emit = emit
    self.timer = None
    if seconds and seconds > 0:
      self.timer = threading.Timer(seconds, self._expire)
      self.timer.daemon = True
      self.timer.start()

  def _expire(self):
    with self.lock:
      if self.done:
        return
      self.done = True
      status = 0
      try:
        self.emit()
      except BaseException:  # noqa: BLE001  (nothing may keep this thread from ending the process)
        import traceback
        traceback.print_exc()
        status = 1  # the line could not be printed: do not look like a success
      finally:
        try:
          sys.stdout.flush()
          sys.stderr.flush()
        finally:
          os._exit(status)

  def cancel(self):
    """True when the caller may go on and print the line itself (the deadline did not fire)."""
    with self.lock:
      if self.done:
        return False
      self.done = True
    if self.timer is not None:
      self.timer.cancel()
    return True


def main():
  global SEPARATE_ROWS
  args = parse()
  SEPARATE_ROWS = not args.slab_rows
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    # `python bench.py --gpus N`: become N ranks, one per GPU (torch.distributed.run, RCCL over xGMI)
    import socket
    with socket.socket() as s:
      s.bind(("127.0.0.1", 0))
      port = s.getsockname()[1]
    os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                              f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
                              str(port), str(ROOT / "bench.py"),
                              *("--dim" + a[3:] if a == "--d" or a.startswith("--d=") else a for a in sys.argv[1:])])
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus and "WORLD_SIZE" in os.environ and args.gpus != 1:
    raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
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
    world = dist.get_world_size()  # the ranks RCCL actually sees
  from byzantinemomentum_amd import build as bm_build
  if rank == 0:
    bm_build.build()  # no-op when the in-tree libbm_gar.so is current
  if distributed:
    dist.barrier()
  import byzantinemomentum_amd as bm
  from byzantinemomentum_amd.sharded import ShardedAggregator, shard_bounds
  bm._lib.load()

  workload = args.workload
  # colwise has no exchange step: its shards are independent units, every rank gets the one-GPU workload ("weak");
  # the distance-based rules and the step exchange statistics: a fixed vector split across the ranks ("strong")
  scaling = args.scaling or ("weak" if (args.weak or workload == "colwise") else "strong")
  base_d = args.d or (D_WRN if workload == "step" else D_RESNET18)
  if scaling == "weak":
    d_total, d = base_d * world, base_d  # every rank: base_d coordinates of a (base_d x world)-coordinate vector
  else:
    d_total = base_d
    lo, hi = shard_bounds(d_total, world, rank)
    d = hi - lo  # this rank's coordinates
  # colwise (the default) has no exchange step: its aggregator — with the library's own RCCL communicator — is only
  # needed by the exchange legs AFTER the timed headline, and is created there, under a deadline
  agg = None if workload == "colwise" else make_aggregator(bm, dist if distributed else None, device, world, rank, distributed)
  timer = KernelTimer()
  per_gar = {}
  extra = {}

  def barrier():
    if distributed:
      dist.barrier()
    torch.cuda.synchronize()

  if workload == "colwise":
    n, f = 25, 5
    stacks = make_stacks(n, f, d, device, 2, 1234 + rank, args.aliased_byz)
    # the unit of the metric is one aggregation of an n x 11.2 M stack: with weak scaling every rank completes two per step
    units = d_total // base_d if scaling == "weak" else 1
    aggs_per_step = 2 * units
    algo_bytes = {"median": 4 * d_total * (n + 1), "trmean": 4 * d_total * (n + 1)}

    def step(i, timed):
      st = stacks[i & 1]
      if timed:
        timer.run("median", lambda: bm.median(st))
        timer.run("trmean", lambda: bm.trmean(st, f))
      else:
        bm.median(st)
        bm.trmean(st, f)
    workload_name = (f"C2 colwise: median + trmean(f={f}), n={n}, total d={d_total}"
                     + (f" = {world} shards of {d} coordinates, one per rank, no collective" if world > 1 else ""))
    dominant_kernel = "colwise_"  # colwise_kernel (plain form) or colwise_burst_kernel
  elif workload == "step":
    from byzantinemomentum_amd.step import AggregationStep
    n, f = 25, 5
    h = n - f
    runner = AggregationStep(n, f, f, gar=args.gar, momentum=0.99, dampening=0.99, attack_factor=1.1, nb_past=25,
                             aggregator=agg)
    gen = torch.Generator(device=device).manual_seed(77 + rank)
    mu_vec = 0.1 * torch.randn(d, device=device, generator=gen)
    # (one allocation per sampled gradient: for the step that placement measured best)
    from byzantinemomentum_amd.sharded import Shards
    sets = [Shards([mu_vec + s * torch.randn(d, device=device, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()],
                   d_total=d_total) for _ in range(2)]
    aggs_per_step = 1
    algo_bytes = {"step": step_algorithmic_bytes(d_total, n, f, args.gar)}

    def step(i, timed):
      if timed:
        timer.run("step", lambda: (runner.run(sets[i & 1]), runner.floats()))
      else:
        runner.run(sets[i & 1])
        runner.floats()
    workload_name = (f"C5 full step mirror (attack.py:757-878): worker momentum 0.99, empire 1.1, rule {args.gar}, "
                     f"study statistics with 25 past gradients; n={n}, f={f}, total d={d_total}")
    dominant_kernel = "momentum_stats_kernel"
  else:
    n, f = (51, 12) if workload == "krum" else (25, 5)
    m = n - f - 2
    from byzantinemomentum_amd.sharded import Shards
    stacks = [Shards(st, d_total=d_total) for st in make_stacks(n, f, d, device, 2, 4321 + rank, args.aliased_byz)]
    aggs_per_step = 1
    algo_bytes = {workload: 4 * d_total * n + 4 * d_total * (m + 1)}
    rule = agg.krum if workload == "krum" else agg.bulyan

    def step(i, timed):
      st = stacks[i & 1]
      bm.gars.invalidate_rank_cache()  # (every call runs its distance pass: see timed_loop)
      if timed:
        timer.run(workload, lambda: rule(st, f))
      else:
        rule(st, f)
    workload_name = (f"{'C3 multi-krum' if workload == 'krum' else 'C4 bulyan'}: n={n}, f={f}, m={m}, total d={d_total}"
                     + (f" dim-sharded over {world} ranks, one all-reduce of the {n}x{n} fp64 partial matrix"
                        if distributed else ""))
    dominant_kernel = "gram3_partial_kernel"

  for i in range(max(args.warmup, 27) if workload == "step" else args.warmup):  # step: fill the deque of 25 pasts
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

  for name, nbytes in algo_bytes.items():
    ms = timer.mean_ms(name)
    if distributed:  # the slowest rank's kernel time
      t = torch.tensor([ms], dtype=torch.float64, device=device)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      ms = t.item()
    per_gar[name] = entry(ms, nbytes, gpus=world, units=aggs_per_step // 2 if workload == "colwise" else 1,
                          config=workload_name)

  def make_line(traffic=None, per_kernel=None):
    """The JSON line from what has been measured so far (rank 0)."""
    dominant = max(algo_bytes, key=lambda k: per_gar[k]["avg_ms"])
    dk = per_gar[dominant]
    if not distributed:
      collectives = "none"
    elif workload == "colwise":
      collectives = "none inside the timed region (the coordinate-wise rules have no exchange step); see `exchange`"
    elif agg is not None and agg.native is not None:
      collectives = "libbm_gar's own RCCL communicator, one C call per aggregation"
    else:
      collectives = "torch.distributed (RCCL)"
    line = {
      "metric": "aggregations/sec (Byzantine-robust GAR over n workers x d dims; achieved HBM GB/s per GAR in roofline/per_gar)",
      "value": aggs_per_step * args.steps / elapsed,
      "unit": "agg/s",
      "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": elapsed / args.steps * 1e3,
      "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
      "dtype": "f32", "data": "synthetic",
      "config": {"workload": workload_name, "n_workers": n, "f": f, "d_total": d_total, "d_per_gpu": d,
                 "byzantine_rows": "aliased" if args.aliased_byz else "distinct buffers",
                 "row_placement": "one torch.empty per row (what a caller of the rules has)" if not args.slab_rows else
                                  "byzantinemomentum_amd.layout.alloc_rows (rows of one allocation, stride = 2 MB multiple + 4352 B) for the C2 / C3 / C4 stacks; one allocation per row for the C5 step",
                 "parallelism": f"dim-shard x{world}" if world > 1 else "single GPU",
                 "collectives": collectives},
      "roofline": {"bound": "hbm", "kernel": dominant, "achieved": dk["gbps"], "peak": HBM_PEAK_GBPS * world,
                   "unit": "GB/s", "frac": dk["gbps"] / (HBM_PEAK_GBPS * world), "traffic": traffic,
                   "traffic_source": None if traffic is None else
                   f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, kernel {dominant_kernel}"
                   f"{' (' + dominant + ')' if per_kernel is not None else ''}: 1024*(2*FETCH_SIZE + WRITE_SIZE) per launch"},
      "per_gar": per_gar,
    }
    if per_kernel is not None:
      line["roofline"]["traffic_per_kernel"] = per_kernel
    line.update(extra)
    return line

  # ---- N > 1: the path's real exchange (sharded Bulyan / Krum with its all-reduce), all-gather, layout exchange ----
  # (also one rank under torchrun with --sharded-extras or --workload bulyan|krum: the same code path, testable on one GPU)
  # The headline above is complete at this point.  What follows creates the library's own RCCL communicator (colwise
  # did not need one) and runs collectives no single-GPU box can exercise with real peers: under a deadline, so that a
  # failure there costs the `exchange` object of the line and not the line.
  cpu_sample = None
  if world == 1 and rank == 0 and not args.no_cpu_baseline and workload in ("bulyan", "krum"):
    cpu_sample = _host_copy(stacks[0])  # (the device stacks go before the extras run)
  exchange_rule = (workload if workload in ("bulyan", "krum") else
                   "bulyan" if workload == "colwise" and not args.no_extras and (world > 1 or args.sharded_extras) else None)
  if distributed and exchange_rule is not None:
    del stacks
    stacks = None
    torch.cuda.empty_cache()
    barrier()  # every rank arms its deadline at about the same moment

    def give_up():
      extra.setdefault("exchange", {})["error"] = (
        f"the exchange legs did not finish within {args.extras_timeout:.0f} s (rank {rank} gave up; a rank stuck in a "
        f"collective, most likely): the legs that did finish are filled in, the others are null.  The headline was "
        f"measured before them and stands.")
      if rank == 0:
        print(json.dumps(make_line()), flush=True)
    deadline = Deadline(args.extras_timeout, give_up)
    try:
      # (tests: BM_BENCH_STALL_S makes this rank hang here, as a rank stuck in a collective would — the deadline must
      #  then print the line; tests/test_gpu_y_bench_exchange.py)
      time.sleep(float(os.environ.get("BM_BENCH_STALL_S", "0") or 0))
      if agg is None:
        agg = make_aggregator(bm, dist, device, world, rank, distributed)
      sharded_extras(bm, agg, dist, device, world, rank, timer, args, exchange_rule,
                     base_d if workload != "colwise" else D_RESNET18, extra, time_rule=(workload == "colwise"), out=per_gar)
    except Exception as err:  # noqa: BLE001
      if deadline.timer is None:
        raise
      import traceback
      traceback.print_exc()
      print(f"[bench rank {rank}] the exchange legs failed: {err!r}", file=sys.stderr, flush=True)
      if deadline.cancel():
        # peers may be inside a collective this rank has left: they end at their own deadlines; this rank ends now
        extra.setdefault("exchange", {})["error"] = f"rank {rank}: {err!r}"
        if rank == 0:
          print(json.dumps(make_line()), flush=True)
        sys.stderr.flush()
        os._exit(0)
      time.sleep(60)  # (the deadline is printing the line and ends the process)
      os._exit(0)
    if not deadline.cancel():
      time.sleep(60)
      os._exit(0)

  # ---- N = 1: the other single-GPU configurations, briefly ----
  if world == 1 and workload == "colwise" and not args.no_extras and rank == 0 and stacks is not None:
    first = stacks[0]
    if not args.no_cpu_baseline:
      extra["cpu_baseline"] = cpu_baseline_colwise(first, f)
    del stacks, first
    torch.cuda.empty_cache()
    per_gar.update(extras_single_gpu(bm, device, timer, args.aliased_byz, cpu_baseline=not args.no_cpu_baseline))
  elif world == 1 and rank == 0 and not args.no_cpu_baseline:
    if workload == "colwise" and stacks is not None:
      extra["cpu_baseline"] = cpu_baseline_colwise(stacks[0], f)
    elif cpu_sample is not None:
      extra["cpu_baseline"] = cpu_baseline_rule(cpu_sample, f, workload)

  if rank == 0:
    dominant = max(algo_bytes, key=lambda k: per_gar[k]["avg_ms"])
    traffic, per_kernel = None, None
    # (not under torch.distributed.run: the child processes would inherit the launcher's rendezvous environment)
    if world == 1 and not distributed and not args.no_traffic and "BM_BENCH_CHILD" not in os.environ:
      child = ["--workload", workload, "--gar", args.gar] + (["--d", str(args.d)] if args.d else []) + \
          (["--aliased-byz"] if args.aliased_byz else [])
      with_extras = workload == "colwise" and not args.no_extras and args.d is None
      if with_extras:  # one pair of passes over the default line with its extras: every dominant kernel at once
        kernels = traffic_kernels(D_RESNET18, D_WRN)
        got = measure_traffic(child, {k: v[0] for k, v in kernels.items()}, extras=True)
        if got is not None:
          per_kernel = {k: {"kernel": kernels[k][0] if isinstance(kernels[k][0], str) else kernels[k][0][0],
                            "traffic": got[k], "algorithmic_bytes": kernels[k][1], "ratio": got[k] / kernels[k][1]}
                        for k in got}
          traffic = got.get(dominant)
      else:
        got = measure_traffic(child, {"dominant": dominant_kernel})
        traffic = None if got is None else got.get("dominant")
    print(json.dumps(make_line(traffic, per_kernel)), flush=True)
  if distributed:
    dist.destroy_process_group()


def step_algorithmic_bytes(d, n, f, gar):
  """4-byte units of d per step as THIS implementation moves them (lists the
  reference's 103 + rule + 3 + 29): fused first pass = sampled + buffers read, buffers written, three
  d-vectors written (ks + 2h + 3 = 63); the rule — for the median / trimmed mean it rides along with the first pass on
  the values that pass holds in registers and costs its output vector only (1); for Krum / Bulyan the distance pass
  rides along (no byte of its own) and the average of the m selected rows / pass 2 remains (m + 1); the study block in one pass (bm_study_stats): sampled avg, honest
  avg, defense, Byzantine vector, newest past, curvature combination C and the past average that leaves the deque
  read, C written (8) — round 2 spent 15 there (attack stats 2, defense stats 1, dots 6, two passes of 3 over C)."""
  h = n - f
  m = n - f - 2
  # Krum / Bulyan: the distance pass rides along with the first pass too (the rows are contracted from its registers);
  # what is left of the rule is the average of the m selected rows / pass 2 over the m ranked rows, + 1 written
  gar_units = {"krum": m + 1, "bulyan": m + 1, "median": 1, "trmean": 1, "phocas": 1, "meamed": 1}.get(gar, n + 1)
  return 4 * d * ((h + 2 * h + 3) + gar_units + 8)


def attack_search(bm, honests, n, f, d, evals=16, gar="krum"):
  """The factor search of the "identical" attacks (attacks/identical.py:67-77, the reference's default
  factor=-16), wall-clock of the whole search in its forms.  Against Multi-Krum (C3): scalar form (one distance
  pass over h+2 rows, then the sixteen candidates from its (h+2)^2 scalars: ON THE DEVICE by default,
  `scalar_form_ms`, the factor never leaves the GPU; on the host after a copy, `host_scalar_form_ms`, what rounds 2-5
  measured) and the reference's form (the rule on the vectors once per evaluation).
  Against the other rules "auto" keeps the exploration's cursor in device memory (bm_search_device_next: the sixteen
  evaluations are queued without a synchronisation; `host_scalar_form_ms` = the same evaluations driven from the host).
  Against the median (C2 shape): every candidate as the middle of (candidate, lo, hi), lo / hi being two order
  statistics of the honest rows formed once per search (the key stays `scalar_form_ms`), and the reference's form.
  Against Bulyan (C4 shape): every candidate ranked from the scalars of one distance pass — by one workgroup from the
  factor in device memory (bm_attack_ranking_device), on the host in `host_scalar_form_ms` —, pass 2 alone on the vectors,
  evaluate only (bm_bulyan_pass2_eval).
  Against the trimmed mean (C2 shape; phocas and meamed alike): every candidate in one pass over the honest rows that
  writes nothing (bm_colwise_eval; the key stays `scalar_form_ms`)."""
  from byzantinemomentum_amd.step import AggregationStep
  avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
  res = {"config": f"empire against {gar}, n={n}, f={f}, d={d}, {evals} evaluations, one GPU"}
  modes = (("auto", 10), ("host", 10), ("generic", 3))
  for mode, reps in modes:
    runner = AggregationStep(n, f, f, gar=gar, attack_evals=evals, line_search=mode, nb_past=0)
    runner._search_factor(honests, avg, direction)
    torch.cuda.synchronize()
    each = []
    for _ in range(reps):
      t0 = time.perf_counter()
      runner.last_factor = runner._search_factor(honests, avg, direction)
      torch.cuda.synchronize()
      each.append((time.perf_counter() - t0) * 1e3)
    factor = runner.last_factor  # (the device form leaves it on the GPU: fetched here, outside the timed calls)
    key = {"auto": "scalar_form", "host": "host_scalar_form", "generic": "per_evaluation_form"}[mode]
    # the MEDIAN search: one search in ten taking 40-60 ms (a stall of the host process, seen on loaded boxes in every
    # round: the same binaries gave a mean of 0.55 ms in one process and 4.7-7.5 ms in the next, with identical legs)
    # would otherwise be the whole figure; the mean, the slowest and every single search ride along
    res[key + "_ms"] = sorted(each)[len(each) // 2]
    res[key + "_mean_ms"] = sum(each) / len(each)
    res[key + "_each_ms"] = [round(v, 4) for v in each]
    res["factor_" + mode] = factor
  res["speedup"] = res["per_evaluation_form_ms"] / res["scalar_form_ms"]
  if gar in ("krum", "bulyan"):
    res["legs"] = search_legs(bm, honests, avg, direction, n, f, gar, evals)
  return res


def search_legs(bm, honests, avg, direction, n, f, gar, evals, reps=10):
  """Where the scalar form of the factor search spends its time, leg by leg (wall clock, each leg bracketed by a
  synchronisation; the entry itself — `scalar_form_ms` — is timed without these brackets): the distance pass over
  h + 2 rows, the copy of the (h+2)^2 fp64 matrix to the host both ways (`.cpu()` into pageable memory / an
  asynchronous copy into a pinned buffer + stream synchronisation: what AggregationStep does), the host arithmetic of
  `evals` candidates.  The same binaries measured 0.55 ms and 6-7 ms per search on one box in round 6, per PROCESS: the
  box facts a slow process would need to explain itself ride along."""
  from byzantinemomentum_amd import linesearch, stats
  h = len(honests)
  k = n - h
  unit = torch.empty_like(avg)
  pinned = torch.empty((h + 2, h + 2), dtype=torch.float64, pin_memory=True)
  where = torch.tensor([0.75], dtype=torch.float64, device=avg.device)
  legs = {"distance_pass_ms": [], "d2h_pageable_ms": [], "d2h_pinned_ms": [], "host_search_ms": [], "device_search_ms": []}
  for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats.multi_fma3([unit], [avg], [direction], 1.0, 1.0)
    sq = bm.gars.pairwise_sqdist(list(honests) + [avg, unit])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ext = sq.cpu().contiguous()
    t2 = time.perf_counter()
    pinned.copy_(sq, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    t3 = time.perf_counter()
    if gar == "krum":
      linesearch.attack_line_search(ext, h, k, f, "krum", evals=evals)
    else:
      for e in range(evals):
        linesearch.attack_ranking(ext, h, k, f, "bulyan", 0.5 + 0.25 * e)
    t4 = time.perf_counter()
    if gar == "krum":  # the same candidates by one workgroup where the matrix is (what line_search="auto" runs)
      stats.attack_search_device(sq, h, k, f, "krum", evals=evals)
      torch.cuda.synchronize()
    else:  # Bulyan: the same rankings by one workgroup each, the factor in device memory (bm_attack_ranking_device)
      for e in range(evals):
        stats.attack_ranking_device(sq, h, k, f, "bulyan", where)
      torch.cuda.synchronize()
    t5 = time.perf_counter()
    for key, dt in zip(legs, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
      legs[key].append(dt * 1e3)
  out = {key: {"median": sorted(v)[len(v) // 2], "max": max(v)} for key, v in legs.items()}
  def read(path):
    try:
      with open(path) as fh:
        return fh.read().strip()
    except OSError:
      return None
  import glob
  stat = read("/proc/self/stat")
  out["box"] = {"cpus_allowed": len(os.sched_getaffinity(0)), "cpu_now": int(stat.split()[38]) if stat else None,
                "governor": read("/sys/devices/system/cpu/cpu0/cpufreq/scaling_governor"),
                "gpu_numa_nodes": sorted({read(p) for p in glob.glob("/sys/class/drm/card*/device/numa_node")} - {None}),
                "numa_nodes": len(glob.glob("/sys/devices/system/node/node[0-9]*")),
                "loadavg": read("/proc/loadavg")}
  return out


def gpu_wake(device, ms=250.0):
  """Keep the GPU busy for `ms` before a timed series that follows a long idle stretch of this process (the CPU baselines
  and the PMC child runs of a default run: 20-60 s).  In 3 of 11 default runs of round 6 the first series after them —
  `krum_c3`, behind its three warm-up calls, 2 ms — ran 10 % slow with one call held up ~40 ms, and nothing after it
  did; the same series without the idle stretch in front never did (profiles/r06_abi23_library_ab.txt).  Untimed."""
  x = torch.empty(1 << 24, device=device)
  t0 = time.perf_counter()
  while (time.perf_counter() - t0) * 1e3 < ms:
    for _ in range(20):
      x.mul_(1.0)
    torch.cuda.synchronize()
  del x


def extras_single_gpu(bm, device, timer, aliased, cpu_baseline=False):
  """C3, C4 (one GPU) and C5 in a few iterations each: the driver-run record then carries every
  single-GPU configuration of BASELINE.json, not only the headline one."""
  from byzantinemomentum_amd.step import AggregationStep
  global SEPARATE_ROWS
  out = {}
  d = D_RESNET18
  gpu_wake(device)
  c3_sample = c4_sample = None
  for name, n, f in (("krum_c3", 51, 12), ("bulyan_c4_1gpu", 25, 5)):
    m = n - f - 2
    stacks = make_stacks(n, f, d, device, 2, 4321, aliased)
    fn = bm.krum if name.startswith("krum") else bm.bulyan
    ms = timed_loop(lambda i: fn(stacks[i & 1], f), 12, 3, timer, name)
    ms_pair = timed_loop(lambda i: bm.gars.pairwise_sqdist(stacks[i & 1]), 12, 3, timer, name + "_dist")
    out[name] = entry(ms, 4 * d * n + 4 * d * (m + 1), config=f"n={n}, f={f}, m={m}, d={d}, one GPU",
                      distance_pass_ms=ms_pair, distance_pass_gbps=4 * d * n / ms_pair / 1e6)
    if name == "krum_c3":
      # (not inside the PMC child run: its per-evaluation form runs the distance kernel on stacks with 12 aliased
      #  candidate rows, which blurs that kernel's per-launch traffic)
      if "BM_BENCH_CHILD" not in os.environ:
        out["attack_search_c3_krum"] = attack_search(bm, stacks[0][:n - f], n, f, d)
      if cpu_baseline:
        c3_sample = _host_copy(stacks[0])  # the full-size host copy, for the CPU baseline at the very end
      # Brute at the same shape: C(51, 12) = 1.6e11 subsets, which the reference's loop (brute.py:47-68) cannot enumerate;
      # bm_brute_select answers from the threshold graphs of the distances
      if "BM_BENCH_CHILD" not in os.environ:  # the same rule on the other row placement
        saved, SEPARATE_ROWS = SEPARATE_ROWS, not SEPARATE_ROWS
        alt = make_stacks(n, f, d, device, 2, 4321, aliased)
        SEPARATE_ROWS = saved
        ms_alt = timed_loop(lambda i: bm.krum(alt[i & 1], f), 12, 3, timer, "krum_alt")
        out["krum_c3_" + ("slab_rows" if saved else "separate_rows")] = entry(
          ms_alt, 4 * d * n + 4 * d * (m + 1), config=f"multi-krum n={n}, f={f}, m={m}, d={d}, rows " + (
            "cut out of one allocation at a skewed stride (layout.alloc_rows)" if saved else "one torch.empty each"))
        del alt
        torch.cuda.empty_cache()
      if "BM_BENCH_CHILD" not in os.environ:
        # classic Krum (m = 1): the same distance pass, then ONE row copied out
        try:
          ms_1 = timed_loop(lambda i: bm.krum(stacks[i & 1], f, 1), 12, 3, timer, "krum_m1")
          out["krum_c3_m1"] = entry(ms_1, 4 * d * n + 4 * d * 2, config=f"classic krum (m=1), n={n}, f={f}, d={d}, one GPU")
        except Exception as err:  # noqa: BLE001  (a side entry must not take the line down)
          out["krum_c3_m1"] = {"error": repr(err)}
      if "BM_BENCH_CHILD" not in os.environ:
        # the reference's largest worker count (reproduce.py:122-162: all six rules at n = 51) at the ResNet-18 length
        for key, fn51, rows_out in (("median_n51", lambda st: bm.median(st), 1), ("trmean_n51", lambda st: bm.trmean(st, f), 1),
                                    ("phocas_n51", lambda st: bm.phocas(st, f), 1), ("meamed_n51", lambda st: bm.meamed(st, f), 1),
                                    ("bulyan_n51", lambda st: bm.bulyan(st, f), None), ("aksel_n51", lambda st: bm.aksel(st, f), None)):
          try:
            ms_51 = timed_loop(lambda i: fn51(stacks[i & 1]), 8, 2, timer, key)
            nbytes = (4 * d * (n + 1) if rows_out == 1 else
                      4 * d * n + 4 * d * (m + 1) if key == "bulyan_n51" else 4 * d * n + 4 * d * ((n + 1) // 2 + 1))
            out[key] = entry(ms_51, nbytes, config=f"{key.split('_')[0]}, n={n}, f={f}, d={d}, one GPU")
          except Exception as err:  # noqa: BLE001  (a side entry must not take the line down)
            out[key] = {"error": repr(err)}
      ms_b = timed_loop(lambda i: bm.brute(stacks[i & 1], f), 12, 3, timer, "brute_c3")
      out["brute_c3"] = entry(ms_b, 4 * d * n + 4 * d * (n - f + 1),
                              config=f"brute.py:32-80, n={n}, f={f} (1.6e11 subsets: not enumerable; the subset of smallest "
                                     f"diameter searched by one workgroup of 16 waves on the device, bm_brute_select_device: no "
                                     f"host round trip), d={d}")
    else:
      if cpu_baseline:
        c4_sample = _host_copy(stacks[0])
      if "BM_BENCH_CHILD" not in os.environ:  # (the PMC child keeps the per-launch traffic of the C2 column kernel clean)
        out["attack_search_c2_median"] = attack_search(bm, stacks[0][:n - f], n, f, d, gar="median")
        # (trimmed mean: every candidate in ONE pass that writes nothing — candidate, rule and objective in registers,
        #  bm_colwise_eval — against candidate vector + rule + objective per evaluation)
        out["attack_search_c2_trmean"] = attack_search(bm, stacks[0][:n - f], n, f, d, gar="trmean")
        try:  # (Bulyan: every candidate ranked on the host from ONE distance pass, only pass 2 on the vectors)
          out["attack_search_c4_bulyan"] = attack_search(bm, stacks[0][:n - f], n, f, d, gar="bulyan")
        except Exception as err:  # noqa: BLE001  (a side entry must not take the line down)
          out["attack_search_c4_bulyan"] = {"error": repr(err)}
      # the other rules of aggregators/ on the C2 / C4 shape (n = 25, f = 5, d = 11.2 M)
      c = (n + 1) // 2
      ms_a = timed_loop(lambda i: bm.aksel(stacks[i & 1], f), 12, 3, timer, "aksel_c2")
      out["aksel_c2"] = entry(ms_a, 4 * d * n + 4 * d * (c + 1),
                              config=f"aksel.py:35-64, n={n}, f={f}, mode mid ({c} rows averaged), d={d}: median fused "
                                     f"with the n row distances, then the selected mean")
      ms_c = timed_loop(lambda i: bm.cge(stacks[i & 1], f), 12, 3, timer, "cge_c2")
      out["cge_c2"] = entry(ms_c, 4 * d * n + 4 * d * (n - f + 1), config=f"cge.py:28-57, n={n}, f={f}, d={d}")
      ms_b = timed_loop(lambda i: bm.brute(stacks[i & 1], f), 12, 3, timer, "brute_n25")
      out["brute_n25"] = entry(ms_b, 4 * d * n + 4 * d * (n - f + 1),
                               config=f"brute.py:32-80, n={n}, f={f} (the first of the 53 130 subsets of smallest diameter, found "
                                      f"by one workgroup on the device without enumerating them, bm_brute_select_device: no host "
                                      f"round trip), d={d}")
      if "BM_BENCH_CHILD" not in os.environ:
        # the headline's column rules on the OTHER row placement, same process (what placement is worth)
        saved, SEPARATE_ROWS = SEPARATE_ROWS, not SEPARATE_ROWS
        alt = make_stacks(n, f, d, device, 2, 1234, aliased)
        SEPARATE_ROWS = saved
        tag = "slab_rows" if saved else "separate_rows"
        how = "cut out of one allocation at a skewed stride (layout.alloc_rows)" if saved else "one torch.empty each"
        for rule, fn in (("median", lambda st: bm.median(st)), ("trmean", lambda st: bm.trmean(st, f))):
          ms_alt = timed_loop(lambda i: fn(alt[i & 1]), 20, 3, timer, rule + "_alt")
          out[f"{rule}_{tag}"] = entry(ms_alt, 4 * d * (n + 1), config=f"{rule}, n={n}, d={d}, rows " + how)
        ms_alt = timed_loop(lambda i: bm.bulyan(alt[i & 1], f), 12, 3, timer, "bulyan_alt")
        out[f"bulyan_c4_1gpu_{tag}"] = entry(ms_alt, 4 * d * n + 4 * d * (m + 1),
                                             config=f"bulyan n={n}, f={f}, m={m}, d={d}, rows " + how)
        del alt
    del stacks
    torch.cuda.empty_cache()
  n, f, d = 25, 5, D_WRN
  h = n - f
  gen = torch.Generator(device=device).manual_seed(77)
  mu_vec = 0.1 * torch.randn(d, device=device, generator=gen)
  # (one allocation per sampled gradient: for the step that placement measured best)
  sets = [[mu_vec + s * torch.randn(d, device=device, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()]
          for _ in range(2)]
  def step_entry(ms, gar, runner):
    """The entry of one step configuration.  Krum / Bulyan: the second pass reads the m selected / ranked rows, of which
    the f_real Byzantine ones are ONE buffer — HBM delivers a buffer once however often the rule counts it — so the
    entry carries both figures: `algorithmic_bytes` row by row as the reference's loop counts them, and
    `distinct_bytes` / `frac_of_8TBps_distinct` with every buffer counted once (what a fraction of the HBM peak can
    honestly be quoted on)."""
    rec = entry(ms, step_algorithmic_bytes(d, n, f, gar), config=f"full step mirror, rule {gar}, n={n}, f={f}, d={d}, one GPU")
    if gar in ("krum", "bulyan") and runner.buffers is not None and runner.last_byzantine is not None:
      rows = list(runner.buffers) + [runner.last_byzantine] * f
      bm.gars.invalidate_rank_cache()
      ranked = (bm.gars.krum_selection(rows, f) if gar == "krum" else bm.gars.bulyan_ranking(rows, f))[:n - f - 2]
      aliased = sum(1 for i in ranked if i >= h)
      distinct = step_algorithmic_bytes(d, n, f, gar) - 4 * d * max(aliased - 1, 0)
      rec.update(selected_aliased_rows=aliased, distinct_bytes=distinct, frac_of_8TBps_distinct=distinct / ms / 1e6 / HBM_PEAK_GBPS)
    return rec

  # (inside the PMC child run only the Krum and Bulyan steps — the median step would launch the C2 column kernel's
  #  instance at another length and blur its per-launch average)
  for gar in (("krum", "bulyan") if "BM_BENCH_CHILD" in os.environ else ("krum", "median")):
    runner = AggregationStep(n, f, f, gar=gar, momentum=0.99, dampening=0.99, attack_factor=1.1, nb_past=25)

    def one(i):
      runner.run(sets[i & 1])
      runner.floats()
    ms = timed_loop(one, 8, 27, timer, "step_" + gar)  # 27 warm-up steps: the deque of 25 past averages is full
    out[f"step_c5_{gar}"] = step_entry(ms, gar, runner)
    del runner
  if "BM_BENCH_CHILD" not in os.environ:
    # the two other rules SURVEY 8d lists for C5: Bulyan (its distance pass rides along with the first pass like
    # Krum's, pass 2 over the 18 ranked rows remains) and the trimmed mean (rides along like the median)
    for gar in ("bulyan", "trmean"):
      try:
        runner = AggregationStep(n, f, f, gar=gar, momentum=0.99, dampening=0.99, attack_factor=1.1, nb_past=25)

        def one_more(i):
          runner.run(sets[i & 1])
          runner.floats()
        ms = timed_loop(one_more, 8, 27, timer, "step_" + gar)
        out[f"step_c5_{gar}"] = step_entry(ms, gar, runner)
        del runner
      except Exception as err:  # noqa: BLE001  (a side entry must not take the line down)
        out[f"step_c5_{gar}"] = {"error": repr(err)}
    # the same step with the momentum at the update, the reference's default placement (attack.py:809-810,837-839): the
    # honest rows are the sampled rows, statistics + Byzantine vector + rule (or its distance pass) are one pass over them
    for gar in ("krum", "median"):
      runner = AggregationStep(n, f, f, gar=gar, momentum=0.99, dampening=0.99, momentum_at="update", attack_factor=1.1,
                               nb_past=25)

      def one_update(i):
        runner.run(sets[i & 1])
        runner.floats()
      ms = timed_loop(one_update, 8, 27, timer, "step_update_" + gar)
      # first pass (+ average of the m selected rows), study block carrying the momentum of the update (M read + written)
      units = (h + 3 if gar == "median" else h + 2 + (n - f - 2) + 1) + 2 + 8
      rec = entry(ms, 4 * d * units,
                  config=f"full step mirror, momentum at the update, rule {gar}, n={n}, f={f}, d={d}, one GPU")
      if gar == "krum" and runner.last_byzantine is not None:
        rows = list(sets[1]) + [runner.last_byzantine] * f  # (the last timed step ran on sets[7 & 1])
        bm.gars.invalidate_rank_cache()
        aliased = sum(1 for i in bm.gars.krum_selection(rows, f) if i >= h)
        distinct = 4 * d * (units - max(aliased - 1, 0))
        rec.update(selected_aliased_rows=aliased, distinct_bytes=distinct, frac_of_8TBps_distinct=distinct / ms / 1e6 / HBM_PEAK_GBPS)
      out[f"step_c5_update_{gar}"] = rec
      del runner
  # last: host threads busy with a baseline must not sit next to a GPU measurement.  Full size, the same stacks.
  if cpu_baseline and c3_sample is not None:
    out["krum_c3"]["cpu_baseline"] = cpu_baseline_rule(c3_sample, 12, "krum")
  if cpu_baseline and c4_sample is not None:
    out["bulyan_c4_1gpu"]["cpu_baseline"] = cpu_baseline_rule(c4_sample, 5, "bulyan")
  del c3_sample, c4_sample
  if cpu_baseline:
    base = cpu_baseline_step(sets[0], n, f, "krum")
    out["step_c5_krum"]["cpu_baseline"] = base
  return out


if __name__ == "__main__":
  main()
