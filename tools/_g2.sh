mkdir -p gpurun_out/r4b
( time python bench.py ) > gpurun_out/r4b/bench_default.json 2> gpurun_out/r4b/bench_default.err
python -m pytest tests -m gpu -x -q > gpurun_out/r4b/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r4b/pytest_gpu.txt
bash tools/pmc_mfma.sh r04 > gpurun_out/r4b/pmc.log 2>&1
tail -3 gpurun_out/r4b/bench_default.err
