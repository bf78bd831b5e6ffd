cd $GRAFT_REPO_ROOT
for dt in bf16x3 bf16 f32; do timeout 300 python tools/host_issue_time.py $dt; done > gpurun_out/g11.txt 2>&1
cat gpurun_out/g11.txt
