mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_c.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r03_gpu_tests_c.log
for cfg in "0 0 12" "127 0 12" "64 64 8" "127 64 8" "0 0 8" "127 32 10"; do
  set -- $cfg
  echo "== RAYCA_TOP_FLAT=$1 RAYCA_TOP_PATH=$2 RAYCA_PATH_LDS_ENTRIES=$3"
  RAYCA_TOP_FLAT=$1 RAYCA_TOP_PATH=$2 RAYCA_PATH_LDS_ENTRIES=$3 timeout -k 10 200 python tests/gpu_ab_inflight.py atrium 4 main 2>&1 | tail -1
done > gpurun_out/r03_ab_top.log 2>&1
cat gpurun_out/r03_ab_top.log
