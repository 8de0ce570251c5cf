# kernel trace of the lln step (GPU box): scripts/step_trace.sh <tag> [exact|reference]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/strace_$1 -o st -- python bench.py --no-cpu-baseline --no-extra-configs --no-other-mode --solver-mode ${2:-exact} --steps 5 --warmup 2 > gpurun_out/strace_$1.log 2>&1 || exit 1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/strace_$1/st_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last 5 steps: find k_rx_count launches as step starts
starts=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith(('k_rx_count', 'k_rx_tiles'))]
seg=rows[starts[-3]:starts[-2]]
t0=int(seg[0]['Start_Timestamp'])
for r in seg:
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    print(f"{(st-t0)/1e3:9.1f} {(en-st)/1e3:8.1f}  {r['Kernel_Name'][:60]}  grid {r.get('Grid_Size','')}")
PY
