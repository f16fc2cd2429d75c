"""Generate tests/golden/study_*.tsv: the 24-column `study` file (attack.py:564-571,870-878) of seeded runs of the
UNMODIFIED reference driver with its OWN rules on CPU (golden vector 2).  torchvision is the
seeded synthetic stand-in of tests/stubs/, so a run is a pure function of the command line below.

    python scripts/make_golden_study.py          # needs the reference checkout (/root/reference or oracle/_ref)

tests/test_oracle_golden.py::test_study_golden_reproduces re-runs the same commands where a checkout is present and
compares; tests/test_gpu_reference_loop.py compares `--gar native-*` on the MI355X with the reference's rule on the
same device (the GPU's backprop differs from the CPU's in the last bits, so the committed files pin the CPU run).
"""
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import reference_loader  # noqa: E402

CASES = {
  "krum_n11_f2_empire_worker": ["--nb-workers", "11", "--nb-decl-byz", "2", "--nb-real-byz", "2", "--gar", "krum",
                                "--attack", "empire", "--attack-args", "factor:1.1", "--momentum-at", "worker"],
  "bulyan_n11_f2_little_update": ["--nb-workers", "11", "--nb-decl-byz", "2", "--nb-real-byz", "2", "--gar", "bulyan",
                                  "--attack", "little", "--attack-args", "factor:1.5", "negative:True",
                                  "--momentum-at", "update"],
  "median_n25_f5_empire_server": ["--nb-workers", "25", "--nb-decl-byz", "5", "--nb-real-byz", "5", "--gar", "median",
                                  "--attack", "empire", "--attack-args", "factor:1.1", "--momentum-at", "server"],
}
COMMON = ["--seed", "1", "--device", "cpu", "--nb-steps", "4", "--model", "simples-full", "--dataset", "mnist",
          "--momentum", "0.9", "--evaluation-delta", "0", "--nb-for-study", "1", "--nb-for-study-past", "3"]


def run_case(args, threads=4):
  """Returns the text of the study file of one run (OMP threads fixed: the CPU matmuls then add in one order)."""
  env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
  env["PYTHONPATH"] = os.pathsep.join([str(ROOT / "tests" / "stubs")] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
  with tempfile.TemporaryDirectory() as tmp:
    out = pathlib.Path(tmp) / "run"
    cmd = [sys.executable, "-OO", os.path.join(reference_loader.REFERENCE_DIR, "attack.py"), *COMMON, *args,
           "--result-directory", str(out)]
    proc = subprocess.run(cmd, env=env, cwd=tmp, capture_output=True, text=True, timeout=600)
    if proc.returncode != 0:
      raise RuntimeError(f"{' '.join(cmd)}\n{proc.stdout[-2000:]}\n{proc.stderr[-2000:]}")
    return (out / "study").read_text()


if __name__ == "__main__":
  if not reference_loader.available():
    raise SystemExit("no reference checkout")
  for name, args in CASES.items():
    text = run_case(args)
    (ROOT / "tests" / "golden" / f"study_{name}.tsv").write_text(text)
    print(name, len(text.splitlines()), "lines")
