#!/bin/bash
# One profiling session on the GPU box: for each bench workload, rocprofv3 kernel stats of bench.py (one frame at a time, so
# that a launch's duration is the kernel's own) and PMC passes.  For the main workload also the trace with frames in flight.
# Summaries land under gpurun_out/; tests/pmc_to_profiles.py copies the judged ones into profiles/.
# The node format (and the soup's camera-ray kernel) is pinned to what the un-profiled bench chooses: under the profiler the
# scene's own timing of the formats is perturbed and can settle on another one.
#   tests/profile_all.sh <session suffix> [workload ...]
export TMPDIR=/tmp
sfx=${1:-r03}; shift
wls=${@:-atrium soup cornell atrium4k}
mkdir -p gpurun_out/prof$sfx
for wl in $wls; do
  case $wl in
    atrium)   export RAYCA_NODE_FORMAT=0; unset RAYCA_REFILL; steps=20; frames=8 ;;
    soup)     export RAYCA_NODE_FORMAT=3 RAYCA_REFILL=1; steps=10; frames=6 ;;
    cornell)  export RAYCA_NODE_FORMAT=1 RAYCA_REFILL=0; steps=40; frames=12 ;;
    atrium4k) unset RAYCA_NODE_FORMAT RAYCA_REFILL; steps=8; frames=4 ;;
  esac
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof$sfx/$wl -- python3 bench.py --workload $wl --no-others --steps $steps --warmup 3 --frames-in-flight 1 --no-cpu-baseline --no-latency > gpurun_out/prof$sfx/bench_$wl.json 2> gpurun_out/prof$sfx/bench_$wl.err || { echo "trace $wl failed"; tail -5 gpurun_out/prof$sfx/bench_$wl.err; exit 1; }
  echo "trace $wl ok"
  if [ $wl = atrium ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof$sfx/atrium_f4 -- python3 bench.py --no-others --steps 20 --warmup 3 --no-cpu-baseline --no-latency > gpurun_out/prof$sfx/bench_atrium_f4.json 2> gpurun_out/prof$sfx/bench_atrium_f4.err || exit 1
  fi
  bash tests/pmc_passes.sh $wl$sfx $wl $frames || exit 1
  python3 tests/pmc_summary.py gpurun_out/pmc_$wl$sfx > gpurun_out/pmc_${wl}${sfx}_summary.txt || exit 1
done
find gpurun_out/prof$sfx -name "*kernel_stats.csv"
