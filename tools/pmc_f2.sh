cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_f2a -o m -- python3 tools/f2_batch_probe.py > gpurun_out/pmc_f2a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_f2b -o l -- python3 tools/f2_batch_probe.py > gpurun_out/pmc_f2b.log 2>&1 || exit 1
