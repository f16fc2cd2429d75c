"""The factor search against Multi-Krum at C3 (n = 51, f = 12, d = 11 173 962) in its three forms, same process, same
rows: candidates evaluated on the device (line_search="auto": bm_attack_line_search_device), on the host after a copy of
the (h+2)^2 matrix ("host": bm_attack_line_search, the form of rounds 2-5), and the rule on the vectors once per
evaluation ("generic", the reference's way) — bench.attack_search, whose legs time the pieces.  Also a whole C5-shaped
step (n = 25, f = 5, d = 36.5 M would not add anything: the search's cost does not depend on d beyond its distance
pass) is NOT run here; `python bench.py` carries the entry as per_gar.attack_search_c3_krum."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import byzantinemomentum_amd as bm  # noqa: E402

dev = torch.device("cuda:0")
n, f, d = 51, 12, bench.D_RESNET18
bench.SEPARATE_ROWS = True
stacks = bench.make_stacks(n, f, d, dev, 1, 4321, False)
honests = stacks[0][:n - f]
for rep in range(int(os.environ.get("REPS", "2"))):
  res = bench.attack_search(bm, honests, n, f, d)
  print(json.dumps(res), flush=True)
  print(f"rep {rep}: device form {res['scalar_form_ms']:.3f} ms (mean {res['scalar_form_mean_ms']:.3f}), host form "
        f"{res['host_scalar_form_ms']:.3f} ms (mean {res['host_scalar_form_mean_ms']:.3f}), per evaluation "
        f"{res['per_evaluation_form_ms']:.2f} ms; factors {res['factor_auto']} {res['factor_host']} {res['factor_generic']}; "
        f"legs " + ", ".join(f"{k} {v['median']:.3f}" for k, v in res["legs"].items() if k != "box"), flush=True)
