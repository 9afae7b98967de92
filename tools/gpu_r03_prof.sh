bash tools/run_profiles.sh r03 > gpurun_out/r03_run_profiles.log 2>&1
cp gpurun_out/r03b/ramp.txt gpurun_out/r03/ 2>/dev/null
tail -5 gpurun_out/r03_run_profiles.log; cat gpurun_out/r03/union_default.txt gpurun_out/r03/union_parts1.txt
