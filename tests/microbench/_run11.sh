mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_d.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03_gpu_tests_d.log
timeout -k 10 500 python bench.py > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err; echo "bench rc=$?"
