# LDS and issue counters of the few-rows kernel (tools/bench_rows_kernel.py, rows given as $1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-128}
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $R/gpurun_out/pmc_rows1 -o a -- python3 $R/tools/bench_rows_kernel.py llama-3-8b 10 $T > /dev/null 2> $R/gpurun_out/pmc_rows1.err
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc_rows2 -o b -- python3 $R/tools/bench_rows_kernel.py llama-3-8b 10 $T > /dev/null 2> $R/gpurun_out/pmc_rows2.err
cd $R
python3 - <<PY
import csv, collections
for f in ("gpurun_out/pmc_rows1/a_counter_collection.csv", "gpurun_out/pmc_rows2/b_counter_collection.csv"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "rows_kernel" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:48], r["Grid_Size"], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(k[0], k[1], k[2], round(v / n))
PY
rm -rf gpurun_out/pmc_rows1 gpurun_out/pmc_rows2
