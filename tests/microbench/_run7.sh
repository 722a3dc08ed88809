mkdir -p gpurun_out
tests/microbench/_build/libm_compare > gpurun_out/r03_libm_compare.jsonl 2> gpurun_out/r03_libm_compare.err; echo "libm rc=$?"
RAYCA_NODE_FORMAT=0 bash tests/pmc_diag_passes.sh atrium_r03 atrium 6 > gpurun_out/r03_diag_atrium.log 2>&1; echo "diag rc=$?"
RAYCA_PROBE_F=3,4 timeout -k 10 300 python tests/gpu_rank_share_probe.py atrium 1 2 4 8 > gpurun_out/r03_rank_share_a.log 2>&1; echo "share rc=$?"
RAYCA_PROBE_F=4 RAYCA_PROBE_CALLS=3 timeout -k 10 300 python tests/gpu_rank_share_probe.py atrium 1 8 > gpurun_out/r03_rank_share_a3.log 2>&1; echo "share3 rc=$?"
RAYCA_PROBE_F=3,4 timeout -k 10 300 python tests/gpu_rank_share_probe.py atrium4k 1 2 4 8 > gpurun_out/r03_rank_share_4k.log 2>&1; echo "share4k rc=$?"
