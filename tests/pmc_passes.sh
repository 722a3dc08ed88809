#!/bin/bash
# separate rocprofv3 --pmc passes (never combined with sys/hip/hsa traces), CSV into gpurun_out/pmc_<tag>/
#   tests/pmc_passes.sh <tag> <workload> <frames>     (RAYCA_NODE_FORMAT / RAYCA_REFILL pinned by the caller)
tag=$1; shift
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}/pass$i -- python3 tests/profile_run.py "$@" > gpurun_out/pmc_${tag}_pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc_${tag}_pass$i.log; exit 1; }
  echo "pass $i ok"
done
