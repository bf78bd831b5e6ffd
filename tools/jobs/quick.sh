# quick check after a kernel edit: op tests, step parity, one bench line, the per-entry table
export TMPDIR=/tmp
mkdir -p gpurun_out/q
(python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -x -q -n 4 2>&1 | tail -4) > gpurun_out/q/tests.log 2>&1
tail -3 gpurun_out/q/tests.log
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
GHM_PROFILE_ALL=1 python bench.py --steps 5 --profile --no-cpu-baseline --no-secondary > /dev/null 2> gpurun_out/q/kernel_table.txt
head -70 gpurun_out/q/kernel_table.txt | cut -c1-110
grep -n "smallk_dgrad\|direct_smallr\| loss \|recon" gpurun_out/q/kernel_table.txt | cut -c1-120
