#!/bin/bash
# The two ceilings on the GPU box: plain run, rocprofv3 kernel trace, and PMC passes (separate from the trace).
# Output under gpurun_out/peaks_<tag>/; tests/microbench/peaks_to_profiles.py copies the judged summary into profiles/.
tag=${1:-r03}
export TMPDIR=/tmp
out=gpurun_out/peaks_$tag
mkdir -p $out &&
tests/microbench/_build/peaks all > $out/peaks.jsonl 2> $out/peaks.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- tests/microbench/_build/peaks all > $out/peaks_under_trace.jsonl 2> $out/trace.err &&
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU --output-format csv -d $out/pmc_valu -- tests/microbench/_build/peaks valu > $out/peaks_pmc_valu.jsonl 2> $out/pmc_valu.err &&
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_l1 -- tests/microbench/_build/peaks l1 > $out/peaks_pmc_l1.jsonl 2> $out/pmc_l1.err &&
timeout -k 10 300 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $out/pmc_l1b -- tests/microbench/_build/peaks l1 > $out/peaks_pmc_l1b.jsonl 2> $out/pmc_l1b.err
echo "peaks rc=$?"
find $out -name "*.csv" | head -20
