mkdir -p gpurun_out
timeout -k 10 400 python tests/gpu_ab_inflight.py atrium 4 main nofilter2 recomp > gpurun_out/r03_ab_filter2.log 2>&1; tail -3 gpurun_out/r03_ab_filter2.log
