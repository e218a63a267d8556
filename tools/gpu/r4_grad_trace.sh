# round 4: kernel timeline of the configs[4] gradient leg (where does the backward's wall time go beyond its kernels?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/grad_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/grad_trace -- python3 tools/bench_grad.py --reps 2 --budget-gb 64 > gpurun_out/grad_trace/out.json 2> gpurun_out/grad_trace/err.log || { tail -5 gpurun_out/grad_trace/err.log; exit 1; }
tail -1 gpurun_out/grad_trace/out.json | cut -c1-400
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/grad_trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
out=[]
for r in rows:
    out.append(((int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-t0)/1e6,r['Kernel_Name'][:60]))
# keep a compact timeline of the LAST second
end=out[-1][1]
with open('gpurun_out/grad_timeline.txt','w') as fo:
    for s,e,n in out:
        if s>end-900: fo.write('%10.3f %10.3f %8.3f %s\n'%(s,e,e-s,n))
print('kernels',len(out),'span ms',end)
PY
rm -rf gpurun_out/grad_trace
