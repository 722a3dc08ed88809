mkdir -p gpurun_out
python -m pytest tests/test_gpu_multi.py tests/test_gpu_engines.py -m gpu -x -q > gpurun_out/r03_gpu_tests_b.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r03_gpu_tests_b.log
bash tests/microbench/run_peaks.sh r03b > gpurun_out/r03_peaks_b.log 2>&1; echo "peaks rc=$?"
