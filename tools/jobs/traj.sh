cd $GRAFT_REPO_ROOT
python tools/trajectory_agreement.py 40 f32 bf16x3 > gpurun_out/r05_trajectory_agreement.txt 2>&1
python tools/trajectory_agreement.py 40 f32 bf16x2 2>&1 | tail -6 >> gpurun_out/r05_trajectory_agreement.txt
python tools/trajectory_agreement.py 40 f32 bf16 2>&1 | tail -6 >> gpurun_out/r05_trajectory_agreement.txt
python tools/trajectory_agreement.py 40 f32 f32 2>&1 | tail -6 >> gpurun_out/r05_trajectory_agreement.txt
tail -45 gpurun_out/r05_trajectory_agreement.txt
