cd $GRAFT_REPO_ROOT
timeout 600 python tools/step_time_series.py --windows 40 --sleep-after 5 > gpurun_out/g8_series.txt 2>&1
timeout 600 python tools/step_time_series.py --windows 12 --dtype f32 >> gpurun_out/g8_series.txt 2>&1
cat gpurun_out/g8_series.txt
