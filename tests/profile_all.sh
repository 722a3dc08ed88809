export TMPDIR=/tmp
mkdir -p gpurun_out/prof6 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6/atrium -- python3 bench.py --steps 20 --warmup 3 > gpurun_out/prof6/bench_atrium.json 2> gpurun_out/prof6/bench_atrium.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6/soup -- python3 bench.py --workload soup --steps 10 --warmup 2 > gpurun_out/prof6/bench_soup.json 2> gpurun_out/prof6/bench_soup.err &&
bash tests/pmc_passes.sh atrium6 atrium pt1 5 &&
bash tests/pmc_passes.sh soup6 soup flat 3 &&
python3 tests/pmc_summary.py gpurun_out/pmc_atrium6 > gpurun_out/pmc_atrium6_summary.txt &&
python3 tests/pmc_summary.py gpurun_out/pmc_soup6 > gpurun_out/pmc_soup6_summary.txt &&
find gpurun_out/prof6 -name "*kernel_stats.csv" | head
