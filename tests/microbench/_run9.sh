mkdir -p gpurun_out
timeout -k 10 300 python tests/gpu_ab_inflight.py atrium 4 prev notop main > gpurun_out/r03_ab_top2.log 2>&1; tail -3 gpurun_out/r03_ab_top2.log
echo "== top path 64, entries 8"
RAYCA_PATH_LDS_ENTRIES=8 timeout -k 10 300 python tests/gpu_ab_inflight.py atrium 4 prev notop main > gpurun_out/r03_ab_top3.log 2>&1; tail -3 gpurun_out/r03_ab_top3.log
