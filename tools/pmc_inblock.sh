#!/bin/bash
# PMC passes over the batched step (128 cells, N = 4096): where do the in-block kernels (rows, short-K updates, diagonal) spend their cycles?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
out=gpurun_out/pmc_inblock
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "FETCH_SIZE WRITE_SIZE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o c -- python3 tools/batch_prof.py 4096 128 1 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 tools/pmc_table.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
rm -rf $out/p*/  # keep only logs + summary
