"""bench.py's command line on a machine without a GPU: the self-spawn for --gpus N, the refusal to run
without a GPU, and the byte model of the step."""

import importlib.util
import os
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def load_bench():
  spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_gpus_flag_reexecutes_under_torch_distributed_run(monkeypatch):
  bench = load_bench()
  calls = []

  def fake_execv(exe, argv):
    calls.append((exe, argv))
    raise SystemExit(0)
  monkeypatch.setattr(os, "execv", fake_execv)
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
  with pytest.raises(SystemExit):
    bench.main()
  (exe, argv), = calls
  assert exe == sys.executable
  assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
  assert "--nproc-per-node=4" in argv and "127.0.0.1" in argv
  tail = argv[argv.index(str(ROOT / "bench.py")) + 1:]
  assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"]     # the ranks see the same flags


def test_launcher_and_flag_must_agree(monkeypatch):
  bench = load_bench()
  monkeypatch.setenv("WORLD_SIZE", "2")
  monkeypatch.setenv("RANK", "0")
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
  with pytest.raises(SystemExit) as err:
    bench.main()
  assert "--gpus 4" in str(err.value)


def test_no_gpu_is_refused_loudly(monkeypatch):
  import torch
  if torch.cuda.is_available():
    pytest.skip("a GPU is visible")
  bench = load_bench()
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  monkeypatch.setattr(sys, "argv", ["bench.py"])
  with pytest.raises(SystemExit) as err:
    bench.main()
  assert "no GPU" in str(err.value)


def test_step_byte_model():
  bench = load_bench()
  d, n, f = 1000, 25, 5
  h, m = 20, 18
  assert bench.step_algorithmic_bytes(d, n, f, "krum") == 4 * d * ((3 * h + 3) + (m + 1) + 8)  # the distance pass rides along
  assert bench.step_algorithmic_bytes(d, n, f, "median") == 4 * d * ((3 * h + 3) + 1 + 8)  # the rule rides along
  assert bench.entry(2.0, 4_000_000_000)["gbps"] == pytest.approx(2000.0)


def test_deadline_prints_what_it_has_and_ends_the_process():
  """A rank stuck in a collective after the timed headline: the deadline's thread prints the line built so far and the
  process leaves with status 0 (bench.Deadline; the main thread never returns from its blocking call)."""
  import subprocess
  code = (
    "import importlib.util, sys, time\n"
    f"spec = importlib.util.spec_from_file_location('b', r'{ROOT / 'bench.py'}')\n"
    "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
    "d = b.Deadline(0.3, lambda: print('{\"value\": 1}'))\n"
    "time.sleep(60)\n"           # stands for a call that never returns (it releases the GIL, as RCCL / ctypes do)
    "print('never')\n")
  done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
  assert done.returncode == 0, done.stderr
  assert done.stdout.strip().splitlines() == ['{"value": 1}']


def test_deadline_cancelled_in_time_does_nothing():
  import time
  bench = load_bench()
  fired = []
  d = bench.Deadline(0.5, lambda: fired.append(1))
  assert d.cancel() is True
  time.sleep(0.8)
  assert fired == []
  assert bench.Deadline(0, lambda: fired.append(2)).cancel() is True and fired == []  # 0 = no limit


def test_cpu_baseline_legs_on_a_small_stack():
  """The three cpu_baseline legs of the line (the reference's own aggregators when its checkout is here, else the
  oracle's pinned port; the loop-body restatement for the step) on a small stack: the records carry value, unit, cores,
  kind, the sample description and the host facts BASELINE.md asks for."""
  import torch
  from oracle import gar_oracle as O
  bench = load_bench()
  threads = torch.get_num_threads()
  try:
    rows, h = O.make_stack("hetero", 25, 5, 4096, seed=99)
    col = bench.cpu_baseline_colwise(rows, 5)
    krum = bench.cpu_baseline_rule(rows, 5, "krum")
    bul = bench.cpu_baseline_rule(rows, 5, "bulyan")
    step = bench.cpu_baseline_step(rows[:h], 25, 5, "krum")
  finally:
    torch.set_num_threads(threads)
  for rec, unit in ((col, "agg/s"), (krum, "agg/s"), (bul, "agg/s"), (step, "steps/s")):
    assert rec["value"] > 0 and rec["unit"] == unit and rec["cores"] >= 1 and rec["kind"] in ("reference", "port")
    assert "full size" in rec["sample"] or "full-size" in rec["sample"]
    assert rec["host"]["hardware_threads"] == os.cpu_count() and rec["host"]["torch"] == torch.__version__
  from oracle import reference_loader
  assert col["kind"] == ("reference" if reference_loader.available() else "port")


def test_self_spawn_spells_the_length_flag_so_that_the_launcher_leaves_it_alone(monkeypatch):
  """`--d N` must reach the ranks as `--dim N`: torch.distributed.run's own parser takes a bare --d for an abbreviation
  of its --duplicate-* options and refuses the command line."""
  bench = load_bench()
  calls = []

  def fake_execv(exe, argv):
    calls.append(argv)
    raise SystemExit(0)
  monkeypatch.setattr(os, "execv", fake_execv)
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  for spelled in (["--d", "1000"], ["--d=1000"], ["--dim", "1000"]):
    calls.clear()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", *spelled])
    with pytest.raises(SystemExit):
      bench.main()
    tail = calls[0][calls[0].index(str(ROOT / "bench.py")) + 1:]
    assert "--d" not in tail and not any(a.startswith("--d=") for a in tail), tail
    assert bench.parse.__call__ is not None
    monkeypatch.setattr(sys, "argv", ["bench.py", *tail])
    assert bench.parse().d == 1000 and bench.parse().gpus == 2


def test_entries_name_their_held_up_calls():
  """per_gar records: mean and median ride along, and a record whose mean a held-up call pulled away from its median
  says so (`held_up_calls`, `timed_calls`, `slowest_ms`) — a record without one carries none of the three."""
  bench = load_bench()

  class FakeEvent:
    def __init__(self, ms):
      self.ms = ms

    def elapsed_time(self, other):
      return other.ms

  timer = bench.KernelTimer()
  timer.pairs["calm"] = [(FakeEvent(0.0), FakeEvent(v)) for v in (0.70, 0.69, 0.71, 0.70, 0.72, 0.68)]
  timer.pairs["stalled"] = [(FakeEvent(0.0), FakeEvent(v)) for v in (0.70, 0.69, 40.0, 0.70, 0.72, 0.68, 0.71, 0.69, 0.70, 0.73, 0.70, 0.69)]
  calm = bench.entry(timer.mean_ms("calm"), 4 * 10**9)
  assert abs(calm["avg_ms"] - 0.70) < 1e-9 and calm["median_ms"] == 0.70 and "held_up_calls" not in calm and "slowest_ms" not in calm
  stalled = bench.entry(timer.mean_ms("stalled"), 4 * 10**9)
  assert stalled["held_up_calls"] == 1 and stalled["timed_calls"] == 12 and stalled["slowest_ms"] == 40.0
  assert stalled["median_ms"] == 0.70 and stalled["avg_ms"] > 3.9
  assert stalled["frac_of_8TBps_median"] > 0.7 > 0.2 > stalled["frac_of_8TBps"]
