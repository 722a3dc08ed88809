mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_g.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03_gpu_tests_g.log
timeout -k 10 300 python tests/gpu_ab_inflight.py atrium 4 prev2 main > gpurun_out/r03_ab_tri64_atrium.log 2>&1; tail -2 gpurun_out/r03_ab_tri64_atrium.log
timeout -k 10 300 python tests/gpu_ab_inflight.py soup 4 prev2 main > gpurun_out/r03_ab_tri64_soup.log 2>&1; tail -2 gpurun_out/r03_ab_tri64_soup.log
