# Bench line, kernel statistics and counter passes of one round (run on the GPU box from the repository root):
#   bash scripts/profile_round.sh   -> gpurun_out/r1i_*  (copy what is to be kept into profiles/)
set -e
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py > gpurun_out/r1i_bench.json 2> gpurun_out/r1i_bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r1i_stats_c2 --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-large > gpurun_out/r1i_stats_c2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r1i_stats_large --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline > gpurun_out/r1i_stats_large.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r1i_pmc_fetch --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-large > gpurun_out/r1i_pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r1i_pmc_write --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --no-large > gpurun_out/r1i_pmc_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/r1i_pmc_f64 --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --steps 5 > gpurun_out/r1i_pmc_f64.log 2>&1
echo done
