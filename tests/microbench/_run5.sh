mkdir -p gpurun_out
(time timeout -k 10 500 python bench.py) > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo "bench rc=$?"; tail -5 gpurun_out/r03_bench_a.err
RAYCA_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 4 --no-others > gpurun_out/r03_bench_n2_gloo.json 2> gpurun_out/r03_bench_n2_gloo.err; echo "n2 rc=$?"; tail -3 gpurun_out/r03_bench_n2_gloo.err
