cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/pmc_fa1 -o a -- python3 $R/tools/bench_fa.py > /dev/null 2> $R/gpurun_out/pmc_fa1.err
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_fa2 -o b -- python3 $R/tools/bench_fa.py > /dev/null 2> $R/gpurun_out/pmc_fa2.err
