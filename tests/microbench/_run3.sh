mkdir -p gpurun_out
tests/microbench/_build/peaks valu > gpurun_out/r03_peaks_c.jsonl 2> gpurun_out/r03_peaks_c.err; echo "rc=$?"
