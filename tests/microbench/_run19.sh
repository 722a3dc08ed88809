mkdir -p gpurun_out
timeout -k 10 400 python tests/gpu_ab_inflight.py atrium 4 prev2 main tri48 > gpurun_out/r03_ab_tri48_atrium.log 2>&1; tail -3 gpurun_out/r03_ab_tri48_atrium.log
timeout -k 10 400 python tests/gpu_ab_inflight.py soup 4 prev2 main tri48 > gpurun_out/r03_ab_tri48_soup.log 2>&1; tail -3 gpurun_out/r03_ab_tri48_soup.log
