# fp64-MFMA utilisation of the batched bench's kernels (separate PMC pass, no tracing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_mfma -o m -- python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only > gpurun_out/pmc_mfma.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d gpurun_out/pmc_lds -o l -- python3 bench.py --steps 3 --warmup 1 --no-extras --batched-only > gpurun_out/pmc_lds.log 2>&1 || exit 1
