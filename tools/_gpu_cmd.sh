bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round2.log 2>&1
