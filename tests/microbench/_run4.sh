mkdir -p gpurun_out
timeout -k 10 400 python tests/gpu_ab_inflight.py atrium 4 main e64 > gpurun_out/r03_ab_e64_atrium.log 2>&1; echo "rc=$?"; cat gpurun_out/r03_ab_e64_atrium.log | tail -4
timeout -k 10 300 python tests/gpu_ab_inflight.py soup 4 main e64 > gpurun_out/r03_ab_e64_soup.log 2>&1; echo "rc=$?"; cat gpurun_out/r03_ab_e64_soup.log | tail -4
