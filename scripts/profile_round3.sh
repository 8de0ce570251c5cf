# Kernel statistics and counter passes of the bench line, written as r3_* (run on the GPU box from the repository root):
#   bash scripts/profile_round3.sh [extra bench.py flags]  -> gpurun_out/r3_*
# scripts/counters_json.py turns the passes into profiles/r3_counters.json; copy what is to be kept into profiles/.
# One pass per counter group, each a run of its own (rocprofv3 does not split counters over passes).
set -e
cd /root/repo
export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-config2 --no-other-mode --no-extra-configs $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3_stats --output-format csv -- $B --steps 10 > gpurun_out/r3_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r3_pmc_fetch --output-format csv -- $B --steps 2 --warmup 1 > gpurun_out/r3_pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r3_pmc_write --output-format csv -- $B --steps 2 --warmup 1 > gpurun_out/r3_pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/r3_pmc_sq --output-format csv -- $B --steps 2 --warmup 1 > gpurun_out/r3_pmc_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES -d gpurun_out/r3_pmc_f64 --output-format csv -- $B --steps 2 --warmup 1 > gpurun_out/r3_pmc_f64.log 2>&1
for d in r3_stats r3_pmc_fetch r3_pmc_write r3_pmc_sq r3_pmc_f64; do
  f=$(find gpurun_out/$d -name "*.csv" | grep -E "kernel_stats|counter_collection" | tail -1)
  cp "$f" gpurun_out/$d.csv
done
python3 scripts/counters_json.py gpurun_out/r3_pmc_fetch.csv gpurun_out/r3_pmc_write.csv gpurun_out/r3_pmc_sq.csv gpurun_out/r3_pmc_f64.csv gpurun_out/r3_counters.json > gpurun_out/r3_counters.log 2>&1
echo done
