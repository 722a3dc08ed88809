mkdir -p gpurun_out
bash tests/profile_all.sh r03b > gpurun_out/r03_profile_b.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/r03_profile_b.log
