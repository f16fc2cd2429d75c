"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean of every counter per dispatch."""
import collections
import csv
import glob
import json
import sys

out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(f"{out}/{tag}_pmc*/**/*counter_collection.csv", recursive=True)):
  with open(path) as fh:
    rows = list(csv.DictReader(fh))
  # one row per (dispatch, counter): sum over dimensions of the same dispatch/counter
  per = collections.defaultdict(float)
  names = {}
  for r in rows:
    key = (r["Dispatch_Id"], r["Counter_Name"])
    per[key] += float(r["Counter_Value"])
    names[r["Dispatch_Id"]] = r["Kernel_Name"]
  for (disp, counter), val in per.items():
    acc[names[disp]][counter].append(val)
summary = {}
for kern, counters in acc.items():
  if not kern.startswith("void bm::") and not kern.startswith("bm::"):
    continue
  summary[kern] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in counters.items()}
json.dump(summary, open(f"{out}/{tag}_pmc_summary.json", "w"), indent=1)
for kern, counters in summary.items():
  print(kern[:100])
  for c, s in sorted(counters.items()):
    print(f"   {c:28s} {s['mean']:.4g}  (n={s['n']})")
