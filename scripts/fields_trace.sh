# kernel trace of the config-5 field step (GPU box): scripts/fields_trace.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 100 python scripts/fields_sweep_probe.py 5 || exit 1
timeout -k 10 100 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ftrace_$1 -o fs -- python scripts/fields_sweep_probe.py 2 > gpurun_out/ftrace_$1.log 2>&1 || exit 1
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/ftrace_$1/fs_kernel_trace.csv')))
for key in ('k_fields<1>','k_fields<2>','k_fields<0>','k_los_trees'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if key in r['Kernel_Name']]
    print(key, [round(x,1) for x in d[-5:]])
PY
