"""bench.py as the driver launches it for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), here with ONE
rank on the one GPU of the box: the exchange legs (libbm_gar's own RCCL communicator, the all-reduce probe, the
all-gather, the all-to-all of the worker-parallel layout) must fill the `exchange` object of the line end to end, and a
deadline that fires on purpose must still leave status 0 and a parsable line.  What makes the first multi-GPU run of
the driver boring.  Collected after the parity files."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(*flags, timeout=420, **environ):
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
  env.update(environ)
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", *flags]
  done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
  lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
  return done, lines


def test_one_rank_launcher_run_fills_the_exchange_object():
  done, lines = _launch("--workload", "bulyan", "--dim", "2000003", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-traffic")
  assert done.returncode == 0, done.stderr[-3000:]
  assert len(lines) == 1, done.stdout[-2000:]
  line = json.loads(lines[0])
  assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0 and line["unit"]
  ex = line["exchange"]
  assert "error" not in ex, ex
  for key in ("ms", "agg_per_s", "allreduce_us", "allreduce_bytes", "allgather_output_ms", "layout_exchange_ms",
              "single_gpu_ms", "speedup_vs_1gpu"):
    if key in ("ms", "agg_per_s", "speedup_vs_1gpu"):  # only timed apart from the headline under --workload colwise
      continue
    assert isinstance(ex[key], (int, float)) and ex[key] > 0, (key, ex)
  assert "libbm_gar" in ex["collectives"], ex["collectives"]   # the library's own communicator was bound, one rank
  assert line["roofline"]["frac"] > 0 and line["config"]["workload"].startswith("C4 bulyan")


def test_a_deadline_that_fires_costs_the_exchange_legs_not_the_line():
  # (the rank hangs for 60 s where the exchange legs start — what a peer lost in a collective looks like — with a
  #  deadline of 2 s: the line must come from the deadline, with the headline and the reason)
  done, lines = _launch("--workload", "bulyan", "--dim", "2000003", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--no-traffic", "--extras-timeout", "2", BM_BENCH_STALL_S="60")
  assert done.returncode == 0, done.stderr[-3000:]
  assert len(lines) == 1, done.stdout[-2000:]
  line = json.loads(lines[0])
  assert line["value"] > 0 and line["ms_per_step"] > 0          # the headline was measured before the legs and stands
  assert "did not finish" in line["exchange"]["error"]
