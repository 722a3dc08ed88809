mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "builder or full_size" > gpurun_out/r03_gpu_tests_e.log 2>&1; echo "builder tests rc=$?"; tail -3 gpurun_out/r03_gpu_tests_e.log
RAYCA_BUILD_LEVELS=1 timeout -k 10 200 python tests/gpu_build_probe.py atrium 5 > gpurun_out/r03_build_probe_a.log 2>&1; echo "probe rc=$?"; grep "probe\]" gpurun_out/r03_build_probe_a.log | tail -6
timeout -k 10 200 python tests/gpu_build_probe.py soup 3 > gpurun_out/r03_build_probe_soup.log 2>&1; grep "probe\]" gpurun_out/r03_build_probe_soup.log | tail -4
