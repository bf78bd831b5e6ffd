#!/bin/bash
# Which unit do the stride-2 split kernels saturate?  (round-5 review, item 4: "260-300 B of global->LDS DMA per wave-MFMA, 3-4x the
# stride-1 kernels" was measured but never tied to a counter.)  One PMC pass per counter group (counters only: --pmc with
# --kernel-trace), the stride-2 kernels beside the stride-1 kernel of the same layer size, each alone (tools/conv_bench.py):
#   gpurun --timeout 1500 -- 'bash tools/pmc_s2_attrib.sh > gpurun_out/r06_pmc_stride2_attribution.txt 2>&1'
export TMPDIR=/tmp
cd /tmp
L=$(rocprofv3 -L 2>/dev/null)
pick() { for c in "$@"; do echo "$L" | grep -qw "$c" && echo -n "$c "; done; }
# texture-addresser / L1 (TCP) / L2 (TCC) side of the DMA path, and the LDS side
G1=$(pick TA_TA_BUSY_sum TA_BUSY_sum TA_TA_BUSY TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum)
G2=$(pick TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum)
G3="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
G4=$(pick SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT)
echo "counter groups available on this box:"; echo "  G1: $G1"; echo "  G2: $G2"; echo "  G3: $G3"; echo "  G4: $G4"
run() {   # geometry, kinds, label
  for gi in 1 2 3 4; do
    eval "CS=\$G$gi"
    [ -z "$CS" ] && continue
    rm -rf /tmp/s2a
    timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/s2a -- python $GRAFT_REPO_ROOT/tools/conv_bench.py $1 --kinds $2 --reps 10 --warm-ms 5 --dtype split --q q > /dev/null 2>&1
    python - "$3" "$gi" <<'P'
import csv, glob, collections, sys
fs = glob.glob('/tmp/s2a/**/*counter_collection.csv', recursive=True)
if not fs:
    print("  (group %s: no counter file)" % sys.argv[2]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    acc[r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:64]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    if not k.startswith('sp_') or 'pack' in k or max(len(x) for x in v.values()) < 5:
        continue
    print("  %-18s G%s %-64s %s" % (sys.argv[1], sys.argv[2], k, "  ".join("%s=%.4g" % (c, sum(x) / len(x)) for c, x in sorted(v.items()))))
P
  done
}
echo "== N8 C64 256x256 -> K128: 3x3 stride 2 (forward, data gradient, weight gradient)"
run "8 64 256 256 128 3 2 1" fwd,dgrad_t,wgrad "s2 C64 256^2 K128"
echo "== N8 C128 128x128 -> K128: 3x3 stride 1 (the same output grid: 128x128 x 128 filters)"
run "8 128 128 128 128 3 1 1" fwd,dgrad_t,wgrad "s1 C128 128^2 K128"
echo "== N8 C128 128x128 -> K256: 3x3 stride 2"
run "8 128 128 128 256 3 2 1" fwd,dgrad_t,wgrad "s2 C128 128^2 K256"
echo "== N8 C256 64x64 -> K256: 3x3 stride 1"
run "8 256 64 64 256 3 1 1" fwd,dgrad_t,wgrad "s1 C256 64^2 K256"
