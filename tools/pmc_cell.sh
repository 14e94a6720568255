#!/bin/bash
# PMC passes (one counter group per run, no tracing) over the one-workgroup-per-cell kernel at N = 1024 x 512 cells
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPRX_CELL_KERNEL=1
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA" "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_cell/$tag -o c -- python3 tools/batch_n1024.py 1024 512 > gpurun_out/pmc_cell_$tag.log 2>&1 || { echo "failed: $set"; tail -3 gpurun_out/pmc_cell_$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_cell/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "potrf_cell_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, sum(v["v"]) / len(v["v"]), len(v["v"]))
PY
