set -e
cd /root/repo
export TMPDIR=/tmp
P="python3 /root/repo/scripts/two_probe.py"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/paint_pmc_a --output-format csv -- $P > gpurun_out/paint_pmc_a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d gpurun_out/paint_pmc_b --output-format csv -- $P > gpurun_out/paint_pmc_b.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU -d gpurun_out/paint_pmc_c --output-format csv -- $P > gpurun_out/paint_pmc_c.log 2>&1
python3 scripts/pmc_summary.py $(find gpurun_out/paint_pmc_a gpurun_out/paint_pmc_b gpurun_out/paint_pmc_c -name "*counter_collection.csv") > gpurun_out/paint_pmc_summary.txt
echo done
