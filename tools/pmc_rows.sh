cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
#  the library is built BEFORE any profiler line; under rocprofv3 a stale library is an error, not a fork + exec of hipcc
python3 -m gpras_amd._build --stale > /dev/null || exit 1
export GPRX_NO_BUILD=1
export GPRX_BATCH_GROUPS=1 GPRX_SPLIT_PANEL=1
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "MemUnitStalled TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_rows/$tag -o c -- python3 tools/batch_probe.py 4096 8 32 > gpurun_out/pmc_rows_$tag.log 2>&1 || exit 1
done
