#!/bin/bash
# In-step A/B of tuning switches: the whole train step (bench.py, no CPU leg) per setting, twice.  Kernels of the two stage
# streams and the gradient stream share the GPU, so a plan that wins in isolation by hiding its own latency (more, smaller
# blocks) can lose in the step, where the other streams' kernels already fill those bubbles -- what counts there is the
# work a kernel costs (issue slots, LDS and HBM bytes), not its stand-alone duration.
#   tools/instep_sweep.sh bf16 "GHM_LP_DGRAD_S2_TILE=0" "GHM_LP_DGRAD_S2_TILE=1" ...
dt=${1:-bf16}; shift
for s in "" "$@"; do
  for r in 1 2; do
    echo -n "${s:-default}: "
    env $s python bench.py --dtype $dt --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
