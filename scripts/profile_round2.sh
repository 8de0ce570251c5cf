# Kernel statistics and counter passes of the bench line, written as r2_* (run on the GPU box from the repository root):
#   bash scripts/profile_round2.sh   -> gpurun_out/r2_*  (scripts/counters_json.py turns the passes into
#   profiles/r2_counters.json; copy what is to be kept into profiles/)
set -e
cd /root/repo
export TMPDIR=/tmp
B="python3 /root/repo/bench.py --no-cpu-baseline --no-config2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2_stats --output-format csv -- $B --steps 10 > gpurun_out/r2_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r2_pmc_fetch --output-format csv -- $B --steps 2 --warmup 1 > gpurun_out/r2_pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r2_pmc_write --output-format csv -- $B --steps 2 --warmup 1 > gpurun_out/r2_pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/r2_pmc_sq --output-format csv -- $B --steps 2 --warmup 1 > gpurun_out/r2_pmc_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d gpurun_out/r2_pmc_f64 --output-format csv -- $B --steps 2 --warmup 1 > gpurun_out/r2_pmc_f64.log 2>&1
find gpurun_out/r2_stats gpurun_out/r2_pmc_fetch gpurun_out/r2_pmc_write gpurun_out/r2_pmc_sq gpurun_out/r2_pmc_f64 -name "*.csv" | head -40
echo done
