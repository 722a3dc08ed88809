#!/bin/bash
# texture-path (TA / TCP / TD) counters of the generation kernel: is the L1 pipe the binding unit?
tag=$1; shift
export TMPDIR=/tmp
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_${tag}/pass$i -- python3 tests/profile_run.py "$@" > gpurun_out/pmc_${tag}_pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/pmc_${tag}_pass$i.log; }
  echo "pass $i done"
done
