#!/bin/bash
# Round 5: PMC passes over the resident Adam loop of the sparse model (16 modes, N = 4096, d = 10, M = 50; 200 steps): ONE counter group per
# run, no tracing with --pmc.  Per kernel: MFMA-busy cycles, fp64 MFMA ops, vector instructions, LDS bank conflicts, wave cycles and waits.
#     bash tools/pmc_sgpr_r5.sh        -> gpurun_out/r05_pmc_sgpr_adam.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
P="python3 tools/sgpr_adam_prof.py 16 200"
D=gpurun_out/pmc_sgpr_r05
rm -rf $D
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $D/p$i -o c -- $P > gpurun_out/r05_pmc_sgpr_p$i.log 2>&1 || { echo "failed: $set"; tail -5 gpurun_out/r05_pmc_sgpr_p$i.log; exit 1; }
done
python3 tools/pmc_table.py $D > gpurun_out/r05_pmc_sgpr_adam.txt
cat gpurun_out/r05_pmc_sgpr_adam.txt
