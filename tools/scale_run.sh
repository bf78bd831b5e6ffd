#!/bin/bash
# The multi-GPU scaling curve of the joint train step on ONE node, for the day a box with more than one MI355X is
# available (the build's GPU boxes expose one): bench.py at 1, 2, 4, 8 ranks, one process per GPU over RCCL / xGMI,
# weak scaling (4 images per GPU).  Each line carries "exchange": RCCL's own rank count, every (sub-)bucket's
# all-reduce time and the exposed wait of each stage stream.
#   tools/scale_run.sh [dtype] [steps]          -> gpurun_out/scale_<dtype>_<N>.json
set -u
cd "$(dirname "$0")/.."
DT=${1:-bf16x3}; STEPS=${2:-30}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
NGPU=$(python -c "from gan_heightmaps_amd import device; print(device.device_count())")
echo "visible GPUs: $NGPU" >&2
for N in 1 2 4 8; do
  [ "$N" -gt "$NGPU" ] && { echo "skip N=$N (only $NGPU GPUs)" >&2; continue; }
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps "$STEPS" --warmup 3 --dtype "$DT" --no-cpu-baseline > "gpurun_out/scale_${DT}_${N}.json"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus "$N" --steps "$STEPS" --warmup 3 --dtype "$DT" > "gpurun_out/scale_${DT}_${N}.json"
  fi
  python - "$N" "gpurun_out/scale_${DT}_${N}.json" <<'PY'
import json, sys
n, path = int(sys.argv[1]), sys.argv[2]
j = json.loads([l for l in open(path) if l.startswith("{")][-1])
x = j.get("exchange") or {}
print("N=%d  %.1f img/s  %.2f ms/step  rccl_nranks=%s  exposed wait %s ms  buckets %s" % (
    n, j["value"], j["ms_per_step"], x.get("rccl_nranks"), x.get("exposed_wait_ms_per_step"),
    [(b["label"], b["MB"], b["avg_ms"]) for b in x.get("buckets", [])]))
PY
done
