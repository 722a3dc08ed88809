mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_f.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_gpu_tests_f.log
(time timeout -k 10 500 python bench.py) > gpurun_out/r03_bench_c.json 2> gpurun_out/r03_bench_c.err; echo "bench rc=$?"; tail -4 gpurun_out/r03_bench_c.err
(time RAYCA_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 4) > gpurun_out/r03_bench_n2_gloo.json 2> gpurun_out/r03_bench_n2_gloo.err; echo "n2 rc=$?"; tail -4 gpurun_out/r03_bench_n2_gloo.err
