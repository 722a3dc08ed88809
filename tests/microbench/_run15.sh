mkdir -p gpurun_out
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_d.json 2> gpurun_out/r03_bench_d.err; echo "bench rc=$?"; tail -4 gpurun_out/r03_bench_d.err
