#!/bin/bash
# One profiling session on the GPU box: rocprofv3 kernel stats of bench.py (one frame at a time, so that a launch's
# duration is the kernel's own; and with the default frames in flight), then PMC passes for atrium and soup.
# Summaries land under gpurun_out/; tests/pmc_to_profiles.py copies the judged ones into profiles/.
# The node format (and the soup's camera-ray kernel) is pinned to what the un-profiled bench chooses: under the profiler the
# scene's own timing of the formats is perturbed and can settle on another one.
export TMPDIR=/tmp
sfx=${1:-r02}
mkdir -p gpurun_out/prof$sfx &&
RAYCA_NODE_FORMAT=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof$sfx/atrium -- python3 bench.py --steps 20 --warmup 3 --frames-in-flight 1 > gpurun_out/prof$sfx/bench_atrium.json 2> gpurun_out/prof$sfx/bench_atrium.err &&
RAYCA_NODE_FORMAT=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof$sfx/atrium_f3 -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof$sfx/bench_atrium_f3.json 2> gpurun_out/prof$sfx/bench_atrium_f3.err &&
RAYCA_NODE_FORMAT=3 RAYCA_REFILL=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof$sfx/soup -- python3 bench.py --workload soup --steps 10 --warmup 2 --frames-in-flight 1 --no-cpu-baseline > gpurun_out/prof$sfx/bench_soup.json 2> gpurun_out/prof$sfx/bench_soup.err &&
RAYCA_NODE_FORMAT=0 bash tests/pmc_passes.sh atrium$sfx atrium pt1 8 &&
RAYCA_NODE_FORMAT=3 RAYCA_REFILL=1 bash tests/pmc_passes.sh soup$sfx soup flat 6 &&
python3 tests/pmc_summary.py gpurun_out/pmc_atrium$sfx > gpurun_out/pmc_atrium${sfx}_summary.txt &&
python3 tests/pmc_summary.py gpurun_out/pmc_soup$sfx > gpurun_out/pmc_soup${sfx}_summary.txt &&
find gpurun_out/prof$sfx -name "*kernel_stats.csv"
