"""Sum rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py <dir> [kernel-substring] -> JSON"""
import csv, glob, json, sys
from collections import defaultdict
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "ionode"
out = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
print(json.dumps({k: dict(v, dispatches=len(n[k])) for k, v in out.items()}, indent=1))
