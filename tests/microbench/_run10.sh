mkdir -p gpurun_out
RAYCA_PROBE_F=3,4 RAYCA_PROBE_COMM_THREAD=1 timeout -k 10 300 python tests/gpu_rank_share_probe.py atrium 1 4 8 > gpurun_out/r03_rank_share_thread.log 2>&1; echo "rc=$?"; grep parts gpurun_out/r03_rank_share_thread.log
