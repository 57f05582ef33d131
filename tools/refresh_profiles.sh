#!/bin/bash
# Runs on the GPU box (from the repository root): the headline bench line, the micro-benchmarks,
# the other model shapes, rocprofv3 kernel statistics of the bench command and the two PMC
# passes for the GEMV's HBM traffic.  Results land under gpurun_out/refresh/; the summaries that
# are kept are copied into profiles/ by tools/keep_refresh.sh.  refresh_profiles.sh PREFIX
# The two PMC passes come FIRST and their result is put where bench.py looks for it
# (profiles/PREFIX_gemv_traffic.json, on the box), so that every bench line of this refresh carries
# the traffic figure of the kernel source it ran.
set -e
RN=${1:?usage: refresh_profiles.sh PREFIX (the kept files become profiles/PREFIX_*)}
R=$(pwd); O=$R/gpurun_out/refresh; mkdir -p $O
B="--no-cpu-baseline --no-prefill-kernels"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 8 --warmup 2 $B > /dev/null 2> $O/pmcf.err
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 8 --warmup 2 $B > /dev/null 2> $O/pmcw.err
cd $R
python tools/collect_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $O/traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
cp $O/traffic.json profiles/${RN}_gemv_traffic.json
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
(timeout -k 10 200 python tools/bench_gemv.py 8b 1b q7 70b && timeout -k 10 120 python tools/bench_attn.py) > $O/microbench.txt 2> $O/micro.err
timeout -k 10 200 python bench.py --model llama-3-70b --context 1024 --steps 64 $B > $O/b70.json 2> /dev/null
timeout -k 10 100 python bench.py --model llama-3.2-1b --context 512 $B > $O/b1.json 2> /dev/null
timeout -k 10 100 python bench.py --model qwen2.5-7b --context 512 $B > $O/bq.json 2> /dev/null
timeout -k 10 100 python bench.py --model stories110M --context 512 $B > $O/bs.json 2> /dev/null
timeout -k 10 100 python bench.py --kv-dtype f16 $B > $O/bkv16.json 2> /dev/null
timeout -k 10 100 python bench.py --kv-dtype f16 --context 8192 $B > $O/bkv16_8k.json 2> /dev/null
timeout -k 10 100 python bench.py --context 8192 $B > $O/b8k.json 2> /dev/null
timeout -k 10 200 python tools/bench_tp_shapes.py > $O/tp.json
timeout -k 10 200 python tools/bench_short_prompts.py > $O/short_prompts.txt 2> /dev/null
timeout -k 10 200 python tools/bench_rows_kernel.py llama-3-8b 60 16,32,64,96,128,192 > $O/few_rows_8b.txt 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err
rm -f $O/prof/b_kernel_trace.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_prefill -o p -- python3 $R/tools/bench_prefill_model.py llama-3-8b 4096 3 > $O/prefill_model.txt 2> $O/prefill_model.err
rm -f $O/prof_prefill/p_kernel_trace.csv
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- python3 $R/tools/bench_prefill.py > $O/prefill_kernels.json 2> $O/pmcm.err
cd $R
python tools/collect_mfma.py $O/pmc_mfma/m_counter_collection.csv $O/prefill_mfma.json > /dev/null
rm -rf $O/pmc_mfma
python tools/summarise_refresh.py $O
