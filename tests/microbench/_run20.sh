mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_h.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_gpu_tests_h.log
bash tests/profile_all.sh r03c > gpurun_out/r03_profile_c.log 2>&1; echo "profile rc=$?"; tail -2 gpurun_out/r03_profile_c.log
