#!/bin/bash
# Regenerates profiles/r01_* on the GPU box (run through gpurun from the repo root); everything lands under
# gpurun_out/refresh/ and is copied into profiles/ afterwards by hand (see profiles/README.md).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 3 > $O/r01_bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 3 --one-stream --no-cpu-baseline > $O/r01_bench_one_stream.json 2>> $O/bench.err
python bench.py --steps 20 --warmup 3 --mode dcgan --no-cpu-baseline > $O/r01_bench_mode_dcgan.json 2>> $O/bench.err
python bench.py --steps 20 --warmup 3 --mode p2p --no-cpu-baseline > $O/r01_bench_mode_p2p.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r01_bench_under_rocprof.json 2>> $O/bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --one-stream > /dev/null 2>> $O/bench.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --one-stream > /dev/null 2>> $O/bench.err
cd $R
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/r01_rocprofv3_kernel_stats.csv \;
python tools/pmc_traffic.py $(dirname $(find $O/pmc_f -name "*counter_collection.csv" | head -1)) $(dirname $(find $O/pmc_w -name "*counter_collection.csv" | head -1)) > $O/r01_pmc_traffic.json
rm -rf $O/stats $O/pmc_f $O/pmc_w
ls -la $O
tail -c 600 $O/r01_bench.json
