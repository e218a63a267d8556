#!/bin/bash
# round 5: per-kernel times of the gradient share (forward with checkpoints, two-phase sweep, reduce)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_grad_trace -- python3 tools/bench_grad.py --reps 1 > gpurun_out/r5_grad_trace.json 2> gpurun_out/r5_grad_trace.err || exit 1
find gpurun_out/r5_grad_trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r5_grad_kernel_stats.csv \;
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r5_grad_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "ionode" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# timeline of the LAST 40 ionode dispatches (one backward pass: recompute / walk / reduce chunks)
for r in rows[-40:]:
    print(r["Kernel_Name"][8:60].ljust(52), round((int(r["Start_Timestamp"]) - t0) / 1e6, 2), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2), r.get("Stream_Id", ""))
PY
find gpurun_out/r5_grad_trace -name "*kernel_trace.csv" -delete
