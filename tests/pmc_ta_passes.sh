#!/bin/bash
# texture-path (TA / TCP / TD) counters of the generation kernel: is the L1 pipe the binding unit?
tag=$1; shift
export TMPDIR=/tmp
i=0
# (the TA_* / TD_* counter sets abort rocprofv3 7.2 on this pool with signal 6: left out)
for set in "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}/pass$i -- python3 tests/profile_run.py "$@" > gpurun_out/pmc_${tag}_pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc_${tag}_pass$i.log; }
  echo "pass $i done"
done
