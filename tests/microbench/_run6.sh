mkdir -p gpurun_out
bash tests/profile_all.sh r03a > gpurun_out/r03_profile_a.log 2>&1; echo "profile rc=$?"; tail -12 gpurun_out/r03_profile_a.log
