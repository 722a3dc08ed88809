mkdir -p gpurun_out
timeout -k 10 300 python tests/gpu_ab_inflight.py atrium 4 main nofilter > gpurun_out/r03_ab_nofilter.log 2>&1; tail -2 gpurun_out/r03_ab_nofilter.log
