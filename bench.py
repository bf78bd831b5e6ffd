#!/usr/bin/env python
"""Throughput of the gan-heightmaps train step on MI355X (BASELINE.json metric:
"512px heightmap+texture train images/sec at 1/2/4/8 MI355X; % MFMA roofline").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one call of the compiled train_fn of experiment test1_nobn_bilin_both (DCGAN G+D and pix2pix
U-Net+PatchGAN forward, four gradient roots, four RMSprop updates) on a synthetic 512x512 batch of 4 per GPU
(weak scaling) that is resident in HBM before the timed region.  Arithmetic: fp32 (the reference's floatX) by operand splitting on the bf16 matrix cores
(dtype "bf16x3").  One process per GPU; gradients are summed
with RCCL.  Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel, HIP events inside the
timed region) and "cpu_baseline" (the numpy oracle on the host cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
LP_MFMA_PEAK_TFLOPS = 2500.0           # MI355X_MICROARCH.md: dense bf16 / fp16 v_mfma_f32_32x32x16 (not the 2:1-sparse 5 PF)
PEAK = {"f32": FP32_MFMA_PEAK_TFLOPS, "bf16": LP_MFMA_PEAK_TFLOPS, "f16": LP_MFMA_PEAK_TFLOPS,
        # fp32 by operand splitting: six bf16 MFMAs per fp32 multiply-add (csrc/conv_split.hip)
        "bf16x3": LP_MFMA_PEAK_TFLOPS / 6.0,
        # two bf16 pieces per operand, three products (BASELINE config 4's arithmetic inside the 1e-3 band)
        "bf16x2": LP_MFMA_PEAK_TFLOPS / 3.0}
JOINT_GFLOP_PER_IMG = 683.29           # algorithmic 2 x MACs of the joint train step
DCGAN_GFLOP_PER_IMG = 391.26           # config 2: DCGAN stage trained (fwd 115.26 + bwd 276.00)
P2P_GFLOP_PER_IMG = 292.03             # config 3: pix2pix stage trained (fwd 95.05 + bwd 196.98)


def synthetic_batch(B, latent_dim, in_shp, seed):
    """Z~U[0,1) (pix2pix.py:31,206); X = uint8/255 (util.py:34); Y = (uint8-127.5)/127.5 (util.py:35)."""
    Z = np.random.RandomState(seed).rand(B, latent_dim).astype(np.float32)
    X = np.random.RandomState(seed + 1).randint(0, 256, (B, 1, in_shp, in_shp)).astype(np.float32) / 255.0
    Y = (np.random.RandomState(seed + 2).randint(0, 256, (B, 3, in_shp, in_shp)).astype(np.float32) - 127.5) / 127.5
    return Z, X.astype(np.float32), Y.astype(np.float32)


def _cpu_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(sample_batch=2, config1_steps=3):
    """The numpy oracle (kind "port": the reference's Theano CPU path cannot be installed here), fp32 im2col + OpenBLAS
    sgemm on all host cores, on a bounded sample: ONE joint train step of the same 512x512 nets at batch
    ``sample_batch`` (>= 2: BatchNorm over a batch of 1 is degenerate), and -- SURVEY 8(d): "img/s at config 1 always"
    -- ``config1_steps`` train steps of BASELINE config 1 (DCGAN 64x64 generator + discriminator, batch 16)."""
    from oracle import step as ostep
    cfg = ostep.default_cfg()
    st = ostep.init_state(cfg, 0, np.float32)
    Z, X, Y = ostep.synthetic_batch(sample_batch, cfg, 0)
    t0 = time.time()
    ostep.train_step(st, Z, X, Y, dtype=np.float32)
    dt = time.time() - t0
    # config 1 (the same geometry tests/test_gpu_step.py::test_train_step_parity[config1_dcgan64_b16] checks on the GPU)
    cfg1 = ostep.default_cfg(in_shp=64, latent_dim=100, train_mode='dcgan',
                             gen_dcgan=dict(nch=64, div=[2, 2, 4, 4]), disc_dcgan=dict(nch=64, div=[8, 4, 2, 1]),
                             gen_p2p=dict(nf=4), disc_p2p=dict(nf=4, mul_factor=[1, 2]))
    st1 = ostep.init_state(cfg1, 0, np.float32)
    Z1, X1, Y1 = ostep.synthetic_batch(16, cfg1, 0)
    ostep.train_step(st1, Z1, X1, Y1, dtype=np.float32)          # warm the BLAS threads
    t1 = time.time()
    for _ in range(config1_steps):
        ostep.train_step(st1, Z1, X1, Y1, dtype=np.float32)
    dt1 = time.time() - t1
    return {"value": sample_batch / dt, "unit": "images/s", "cores": int(_cpu_threads()), "kind": "port",
            "sample": "1 joint train step (fwd + 4 gradient roots + RMSprop) of the same 512x512 "
                      "test1_nobn_bilin_both nets at batch %d, numpy fp32 im2col+sgemm oracle, %.1f s"
                      % (sample_batch, dt),
            "config1": {"value": round(16 * config1_steps / dt1, 2), "unit": "images/s",
                        "sample": "%d DCGAN train steps at 64x64, batch 16 (BASELINE config 1; the p2p nets of the "
                                  "oracle's step are 4-filter stubs, train_mode='dcgan'), %.1f s" % (config1_steps, dt1)}}


MAX_TIMERS = 4096       # csrc/common.h GHM_MAX_TIMERS: recorded timer slots wrap modulo this


SECONDARY = [
    # (name, overrides): BASELINE.json configs 1-5 beside the headline, each a short timed loop of its own AFTER the
    # headline's timed region (never inside it), printed under one "secondary" key of the same JSON line.
    # The headline arithmetic (round-4 review's ruling) is fp32 by operand splitting on the bf16 matrix cores (dtype
    # "bf16x3"); the same workload on v_mfma_f32_32x32x2_f32 stays in the line as the first secondary value
    ("headline_workload_fp32_mfma", dict(dtype="f32")),
    ("config4_per_gpu_bf16x2_512_b4", dict(dtype="bf16x2")),
    ("config4_per_gpu_bf16_512_b4", dict(dtype="bf16")),
    ("config5_per_gpu_f16_1024_b2", dict(dtype="f16", in_shp=1024, batch_per_gpu=2)),
    ("config2_dcgan_512_b4_fp32_by_bf16x3_splitting", dict(mode="dcgan")),
    ("config3_p2p_512_b4_fp32_by_bf16x3_splitting", dict(mode="p2p")),
    ("config1_dcgan64_b16_fp32_by_bf16x3_splitting", dict(config1=True)),
    ("config2_dcgan_512_b4_fp32_mfma", dict(mode="dcgan", dtype="f32")),
    ("config3_p2p_512_b4_fp32_mfma", dict(mode="p2p", dtype="f32")),
]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=4)
    ap.add_argument("--mode", default="both", choices=["both", "dcgan", "p2p"])
    ap.add_argument("--graph", action="store_true", help="replay the step as a captured HIP graph (no per-kernel events)")
    ap.add_argument("--issue", default="recorded", choices=["recorded", "eager"],
                    help="recorded (default): the step's multi-stream launch sequence is recorded once in libghm.so and "
                         "every step is ONE ghm_step_run call; eager: one C call per kernel from Python")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-stream", action="store_true", help="run both GAN stages on a single HIP stream")
    ap.add_argument("--no-grad-streams", action="store_true",
                    help="keep the weight/bias gradients on the stage's own stream (default: a second stream per stage)")
    ap.add_argument("--profile", action="store_true", help="print a per-program-entry timing table to stderr")
    ap.add_argument("--ablate", default="", help="TUNING ONLY (results are wrong): comma-separated program-entry labels or "
                    "kernel-name prefixes whose launches are skipped, to see what a class of kernels costs inside the "
                    "overlapped schedule; the JSON line is marked invalid")
    ap.add_argument("--repeat", default="", help="TUNING ONLY: comma-separated program-entry labels or kernel-name prefixes whose "
                    "launches are issued TWICE: the step's increase is what the class costs inside the overlapped schedule, on "
                    "real data (skipping a producer leaves zeros in matrix-core operands, which run ~18 %% faster); the JSON line "
                    "is marked invalid")
    ap.add_argument("--dtype", default="bf16x3", choices=["f32", "bf16", "f16", "bf16x3", "bf16x2"],
                    help="arithmetic of the convolution products.  bf16x3 (default, the headline) = the reference's "
                         "floatX=float32 arithmetic on the bf16 matrix cores by operand splitting: three bf16 pieces per fp32 "
                         "value that sum to it exactly, six products, fp32 accumulation -- fp32-accurate, held to the fp32 "
                         "path's parity bounds; f32 = the same arithmetic on v_mfma_f32_32x32x2_f32 (a named secondary value); "
                         "bf16 / f16 = BASELINE configs 4 / 5 (matrix-core operands rounded, fp32 accumulation, fp32 tensors / "
                         "master weights / optimiser): additional lines, never the headline; bf16x2 = config 4 with two bf16 "
                         "pieces per operand and three products (outputs inside north_star's 1e-3, half the matrix-core time of bf16x3)")
    ap.add_argument("--config1", action="store_true",
                    help="BASELINE config 1 instead of the headline workload: DCGAN 64x64 generator + discriminator "
                         "(nch 64, div [2,2,4,4] / [8,4,2,1]), batch 16, train_mode='dcgan' -- an extra line for BASELINE.md, "
                         "never the headline")
    ap.add_argument("--in-shp", type=int, default=512, choices=[512, 1024],
                    help="1024 = BASELINE config 5 geometry (one more U-Net level and DCGAN stage; beyond the reference)")
    ap.add_argument("--exchange", default=None, choices=["allreduce", "rs_ag", "allreduce_bf16"],
                    help="N > 1: form of the gradient exchange (default allreduce; rs_ag = reduce-scatter, sharded optimiser "
                         "update, all-gather of the updated parameters; allreduce_bf16 = the all-reduce through a bf16 exchange "
                         "buffer: half the bytes, REDUCED precision -- the JSON line says so in config.exchange)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline only: skip the short extra loops of BASELINE configs 1-5 (\"secondary\" key)")
    ap.add_argument("--secondary-steps", type=int, default=12)
    ap.add_argument("--batches", type=int, default=8,
                    help="distinct synthetic batches resident in HBM that the timed steps rotate through (device-to-device "
                         "into the step's input buffers on a copy stream while the previous step runs); 1 = the same batch "
                         "every step (rounds 1-5; reported as value_single_batch)")
    ap.add_argument("--side-file", default=os.path.join("gpurun_out", "bench_secondary.json"),
                    help="where the secondary configurations / thin-layer table / notes go (the stdout line stays < 4 KB)")
    ap.add_argument("--minimal", action="store_true",
                    help="timed region only: no steady-state loop, no host-array loops (counter-collection runs, where every "
                         "dispatch is serialised and a step takes seconds)")
    return ap.parse_args(argv)


def nominal_flops_per_step(b):
    """ALGORITHMIC 2 x MACs of one train step, summed over the plan's convolution launches (SURVEY 8d prices every layer
    in the reference's own form): an Upscale2D -> 5x5 convolution executes as a collapsed 3x3 with 4K filters (9 MACs per
    output instead of 25), and the first discriminator layer's gradients gather over pooled elements (a quarter of the
    dense products) -- both are counted at the reference's dense figure here."""
    tot = 0.0
    for lane in b.train_compute:
        for e in lane:
            meta = e[2] if len(e) > 2 else None
            if not meta or not meta.get("flops"):
                continue
            f = meta.get("nominal_flops", meta["flops"])     # (a collapsed bilinear convolution executes 25 of its 36 taps)
            if e[0].startswith("upconv"):
                f *= 25.0 / 9.0
            if meta["kernel"].startswith("pool_thin"):
                f *= 4.0
            tot += f
    return tot


def measure(args, secondary_name=None):
    """one workload: build, warm up, time ``args.steps`` steps -> the JSON object of bench.py's contract"""

    from gan_heightmaps_amd import device, dist
    from gan_heightmaps_amd.experiments import make_model

    rank, local_rank, world = dist.env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    if device.device_count() == 0:
        sys.exit("bench.py: no HIP device visible (there is no CPU fallback)")
    ndev = device.device_count()
    if local_rank >= ndev:
        # a launcher that narrows visibility per rank (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES) leaves one device 0
        sys.stderr.write("bench.py: LOCAL_RANK %d but %d visible device(s): using device %d\n"
                         % (local_rank, ndev, local_rank % ndev))
        local_rank %= ndev
    dev = device.Device(local_rank)
    # the communicator gets its own context of the same GPU: its stream is the communication stream, so the bucket
    # all-reduces run beside the rest of the backward pass (step.py)
    cdev = device.Device(local_rank) if world > 1 else None
    comm = dist.Comm(cdev, rank, world) if world > 1 else None
    B = args.batch_per_gpu
    issue = True if args.graph else ('recorded' if args.issue == 'recorded' else False)
    backend = dict(device=dev, comm=comm, use_graph=issue, seed=0, verbose=False, two_streams=not args.one_stream,
                   side_streams=(not args.no_grad_streams) and not args.graph, dtype=args.dtype, exchange_mode=args.exchange)
    S = args.in_shp
    if args.config1:
        # the geometry tests/test_gpu_step.py::test_train_step_parity[config1_dcgan64_b16] checks against the oracle
        from gan_heightmaps_amd.experiments import experiment_kwargs
        from gan_heightmaps_amd.pix2pix import Pix2Pix
        S, B = 64, 16
        kw = experiment_kwargs('test1_nobn_bilin_both')
        kw.update(in_shp=64, latent_dim=100, train_mode='dcgan', **backend)
        kw['gen_params_dcgan'] = {'nch': 64, 'div': [2, 2, 4, 4], 'num_repeats': 0}
        kw['disc_params_dcgan'] = dict(kw['disc_params_dcgan'], nch=64, div=[8, 4, 2, 1])
        kw['gen_params_p2p'] = dict(kw['gen_params_p2p'], nf=4)
        kw['disc_params_p2p'] = dict(kw['disc_params_p2p'], nf=4, mul_factor=[1, 2])
        model = Pix2Pix(**kw)
        args.mode = 'dcgan'
    elif S == 512:
        # configs 2 / 3 of BASELINE.json: the same nets with train_mode='dcgan' / 'p2p' (pix2pix.py:136-141), passed to
        # the constructor as the reference's experiments pass it
        model = make_model('test1_nobn_bilin_both', **backend, **({} if args.mode == 'both' else {'train_mode': args.mode}))
    else:
        # config 5: the same architecture functions one level deeper (p2p.py:137 asserts 512 in the reference)
        from gan_heightmaps_amd.experiments import experiment_kwargs
        from gan_heightmaps_amd.pix2pix import Pix2Pix
        kw = experiment_kwargs('test1_nobn_bilin_both')
        kw.update(in_shp=S, **backend)
        kw['gen_params_dcgan'] = {'num_repeats': 0, 'div': [2, 2, 4, 4, 8, 8, 8, 8], 'final_size': S}
        kw['disc_params_dcgan'] = dict(kw['disc_params_dcgan'], div=[8, 8, 4, 4, 4, 2, 2, 2], nch=1024)
        model = Pix2Pix(**kw)
    if args.mode != 'both' and S != 512:
        model.engine.train_mode = args.mode
    eng = model.engine
    Z, X, Y = synthetic_batch(B, 100 if args.config1 else 1000, S, seed=1000 + rank)
    b = eng.built(B)
    latent = 100 if args.config1 else 1000
    # the timed steps ROTATE through ``args.batches`` distinct synthetic batches that lie in HBM before the timed region: a
    # second plan of the same batch size (own activations and inputs, shared parameters: the product's input pipeline,
    # step.py) takes batch k+1 by device-to-device copies on the copy stream while step k runs on the other plan.  (Rounds
    # 1-5 re-trained ONE batch 85+ times: the discriminator over-fits it, and the split kernels' rate is data-dependent.)
    rotate = args.batches > 1 and issue is not False
    plans = [b] + ([eng.built(B, 1)] if rotate else [])
    for p_ in plans:
        if args.ablate:
            pats = [p for p in args.ablate.split(",") if p]

            def dead(e):
                k = e[2]["kernel"] if len(e) > 2 and e[2] else ""
                return any(e[0] == p or (k and k.startswith(p)) for p in pats)
            for lanes in (p_.train_compute, p_.update):
                for i in (0, 1):
                    lanes[i][:] = [((e[0], (lambda: None)) + tuple(e[2:])) if dead(e) else e for e in lanes[i]]
        if args.repeat:
            rpats = [p for p in args.repeat.split(",") if p]

            def twice(e):
                k = e[2]["kernel"] if len(e) > 2 and e[2] else ""
                return any(e[0] == p or (k and k.startswith(p)) for p in rpats)
            for lanes in (p_.train_compute, p_.update):
                for i in (0, 1):
                    lanes[i][:] = [x for e in lanes[i] for x in ((e, e) if twice(e) else (e,))]
    kstep = [0]                 # steps issued so far: step k runs on plans[k & 1] with batch k % len(pool)
    if rotate:
        pool = [tuple(dev.tensor(a) for a in synthetic_batch(B, latent, S, seed=1000 + 1000 * rank + 3 * i))
                for i in range(args.batches)]           # inputs resident in HBM before the timed region
        eng.upload_resident_async(plans[0], *pool[0])

        def cur_plan():
            return kstep[0] & 1

        def run_steps(n, wrap=None):
            for _ in range(n):
                k = kstep[0]
                eng.enqueue_train_uploaded(plans[k & 1], wrap)
                kstep[0] = k + 1
                eng.upload_resident_async(plans[(k + 1) & 1], *pool[(k + 1) % len(pool)])
    else:
        eng._upload(b, Z, X, Y)                      # inputs resident in HBM before the timed region

        def cur_plan():
            return 0

        def run_steps(n, wrap=None):
            for _ in range(n):
                eng.enqueue_train(b, wrap)
                kstep[0] += 1

    # set-up steps (untimed, not the warm-up): call 0 of a plan is eager (library workspaces take their size), call 1 records
    run_steps(2 * len(plans) if issue == 'recorded' else len(plans))
    eng.sync()

    # ---- pick the dominant kernel from one instrumented (untimed) step ----
    dominant, launches_per_step, flops_per_step = None, 0, 0.0
    executed_flops_per_step = None
    thin_rows = []
    if not args.graph:
        # three instrumented (untimed) steps, per-entry median: every entry runs alone between two HIP events
        runs = [eng.profile_train(B) for _ in range(3)]
        table = [(r[0][0], sorted(x[1] for x in r)[1], r[0][2], r[0][3]) for r in zip(*runs)]
        by_kernel = {}
        for label, ms, meta, _lane in table:
            k = meta["kernel"].split(" splits")[0] if meta else label
            e = by_kernel.setdefault(k, [0.0, 0, 0.0])
            e[0] += ms
            e[1] += 1
            e[2] += meta["flops"] if meta else 0.0
        if args.profile and rank == 0:
            streams = {}
            for r0_, ms_ in zip(runs[0], [t[1] for t in table]):
                if r0_[0] not in ("fork", "join"):
                    streams[r0_[3]] = streams.get(r0_[3], 0.0) + ms_
            print("isolated time per lane (A / B = DCGAN / pix2pix stage stream, ' = that stage's work on the shared gradient stream, C = communication): "
                  + ", ".join("%s %.2f ms" % kv for kv in sorted(streams.items())), file=sys.stderr)
            tot = sum(v[0] for v in by_kernel.values())
            for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][0]):
                print("%-44s %9.3f ms %5.1f%% n=%3d %8.1f GFLOP %6.1f TF/s" %
                      (k, v[0], 100 * v[0] / tot, v[1], v[2] / 1e9, v[2] / 1e9 / max(v[0], 1e-9)), file=sys.stderr)
            for label, ms, meta, lane_ in sorted(table, key=lambda t: -t[1])[:(400 if os.environ.get('GHM_PROFILE_ALL') else 40)]:
                print("  %-2s %-18s %8.3f ms %s %s" % (lane_, label, ms, meta["kernel"] if meta else "",
                                                      meta["geom"] if meta else ""), file=sys.stderr)
        # the thin first / last layers (<= 4 channels on one side) are HBM-bound: reported against their byte floor
        # (input + output tensor once) and the 8 TB/s HBM peak, not against the MFMA peak
        # ... timed WARM: ten launches back to back behind three untimed ones.  (One launch between two events behind a host
        # synchronisation -- the table above -- charges a 20-100 us kernel its launch latency from an idle GPU: 78 against
        # 21 us on the U-Net's first layer.)
        warm = {}
        for lane in (0, 1):
            for e in b.train_compute[lane]:
                if len(e) > 2 and e[2] and e[2].get("thin"):
                    d_ = e[3] if len(e) > 3 and e[3] is not None else eng.devs[lane]
                    eng.sync()
                    for _ in range(3):
                        e[1]()
                    d_.timer_start(1)
                    for _ in range(10):
                        e[1]()
                    d_.timer_stop(1)
                    warm[id(e[2])] = d_.timer_ms(1) / 10
        eng.sync()
        table = [(label, warm.get(id(meta), ms), meta, lane_) for label, ms, meta, lane_ in table]
        thin_rows = [{"entry": label, "kernel": meta["kernel"].split(" splits")[0], "geom": meta["geom"], "ms": round(ms, 4),
                      "algorithmic_MB": round(meta["bytes"] / 1e6, 1), "GB/s": round(meta["bytes"] / ms / 1e6, 1),
                      "frac_of_8TB/s": round(meta["bytes"] / ms / 1e6 / 8000.0, 3),
                      # what the launch really moves (the q copy of its result on top of / instead of the fp32 tensor)
                      "moved_MB": round(meta.get("moved_bytes", meta["bytes"]) / 1e6, 1),
                      "moved_GB/s": round(meta.get("moved_bytes", meta["bytes"]) / ms / 1e6, 1)}
                     for label, ms, meta, _lane in table if meta and meta.get("thin") and ms > 0]
        dominant = max((k for k in by_kernel if by_kernel[k][2] > 0), key=lambda k: by_kernel[k][0])
        launches_per_step = by_kernel[dominant][1]
        flops_per_step = by_kernel[dominant][2]
        executed_flops_per_step = sum(v[2] for v in by_kernel.values())     # MACs the kernels really perform
        isolated_ms = by_kernel[dominant][0] / launches_per_step      # one kernel at a time, nothing else on the GPU

    def is_dom(e):
        return len(e) > 2 and e[2] is not None and e[2]["kernel"].split(" splits")[0] == dominant

    def is_comm(e):          # data-parallel runs: every (sub-)bucket all-reduce and each stage stream's wait for them
        return world > 1 and (e[0].startswith(("allreduce_", "reducescatter_", "allgather_", "rmsprop_shard_", "adam_shard_"))
                              or e[0] == "wait_comm" or e[0].startswith("wait_gather_"))

    slots = []          # device of every bracketed entry, in slot order
    slot_labels = []    # None = a launch of the dominant kernel, else the exchange entry's label

    def wrap(lane, e):
        if is_dom(e) or is_comm(e):
            d = e[3] if len(e) > 3 and e[3] is not None else eng.devs[lane]
            d.timer_start(len(slots))
            e[1]()
            d.timer_stop(len(slots))
            slots.append(d)
            slot_labels.append(None if is_dom(e) else e[0] + ("@A" if d is eng.devs[0] else "@B") * (e[0] == "wait_comm"))
        else:
            e[1]()

    base, nslot, runs_of, stride = {}, {}, {}, 0
    if issue == 'recorded' and dominant:
        # re-record the step of every plan with HIP-event brackets around every launch of the dominant kernel; the recorded
        # timer slots advance by ``stride`` (= the brackets of all plans) per replay, so every launch of the timed region has
        # its own events
        for _ in plans:
            p = cur_plan()
            base[p] = len(slots)
            plans[p].steps.pop('train', None)
            run_steps(1, wrap)                  # records (with the brackets) and replays once: untimed
            nslot[p] = len(slots) - base[p]
            runs_of[p] = 1                      # the recording call was replay 0 of this plan's step
        stride = len(slots)
        for p in base:
            type(dev).step_timer_stride(plans[p].steps['train'], stride)
        rec_devs = list(slots)
        eng.sync()
        inst_steps = args.steps
    else:
        inst_steps = min(args.steps, 4000 // max(launches_per_step + 24 * (world > 1), 1)) if dominant else 0

    # ---- the W untimed warm-up steps, immediately in front of the timed region (the instrumented per-entry steps above leave
    # the GPU idle between launches: clocks and caches are those of an idle chip behind them) ----
    for _ in range(args.warmup):
        if issue is not False:
            runs_of[cur_plan()] = runs_of.get(cur_plan(), 0) + 1       # (a replay: the recorded timer slots advance)
        run_steps(1)
    # ---- timed region ----
    if comm is not None:
        comm.barrier()
    eng.sync()
    order = []                  # (plan, replay number of that plan's recorded step) of every timed step
    t0 = time.perf_counter()
    for s in range(args.steps):
        if issue is not False:
            p = cur_plan()
            order.append((p, runs_of.get(p, 0)))
            runs_of[p] = runs_of.get(p, 0) + 1
            run_steps(1)                        # ONE ghm_step_run (recorded) / two graph launches (--graph)
        else:
            run_steps(1, wrap if s < inst_steps else (lambda lane, e: e[1]()))
    eng.sync()
    if comm is not None:
        comm.barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max_scalar(elapsed)
    losses = eng._read_losses()
    if issue == 'recorded' and dominant:
        # replay r of plan p used slots base[p] + i + r * stride; the slots wrap modulo MAX_TIMERS: the last ``keep`` replays
        # of a plan still hold their events
        keep = MAX_TIMERS // max(stride, 1) - 1
        every = [(rec_devs[i], (i + r * stride) % MAX_TIMERS, slot_labels[i])
                 for p, r in order if runs_of[p] - r <= keep for i in range(base[p], base[p] + nslot[p])]
    else:
        every = [(d, i, slot_labels[i]) for i, d in enumerate(slots)]
    timed = [(d, i) for d, i, lab in every if lab is None]
    slot = len(timed)
    # read every bracket of the timed region NOW: the loops below replay the bracketed steps and the recorded timer slots
    # wrap modulo MAX_TIMERS, so later replays would overwrite these event pairs
    timed_ms = [d.timer_ms(i) for d, i in timed]
    every_ms = {(id(d), i): d.timer_ms(i) for d, i, lab in every if lab is not None}
    extras = (issue is not False and not args.ablate and not args.repeat and world == 1 and not secondary_name
              and not args.minimal)
    # ---- steady state: the split kernels' rate is data- and clock-dependent (the bf16 matrix pipes are power-managed, DESIGN
    # section 4d): 60 more steps behind the timed region, reported beside ``value`` (never instead of it) ----
    steady = None
    if args.dtype == "bf16x3" and extras:
        eng.sync()
        t1 = time.perf_counter()
        run_steps(60)
        eng.sync()
        steady = {"steps": 60, "after_steps": args.steps + args.warmup, "value": round(B * 60 / (time.perf_counter() - t1), 3)}
    # ---- the same steps with lr = 0 (parameters frozen: what the kernels do on a FIXED state of the nets; the timed region
    # above trains, so its activations drift with the parameters) and, as rounds 1-5 measured, on ONE batch repeated ----
    value_lr0 = value_single = None
    if extras:
        eng.sync()
        lr_saved = float(eng.hyper['dcgan_gen'].numpy().ravel()[0])
        eng.set_lr(0.0)
        run_steps(2)
        eng.sync()
        t1 = time.perf_counter()
        run_steps(args.steps)
        eng.sync()
        value_lr0 = B * args.steps / (time.perf_counter() - t1)
        eng.set_lr(lr_saved)
        if rotate:
            eng._upload(b, Z, X, Y)
            for _ in range(2):
                eng.enqueue_train(b)
            eng.sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                eng.enqueue_train(b)
            eng.sync()
            value_single = B * args.steps / (time.perf_counter() - t1)
    # ---- the same loop with the reference's host-array boundary (pix2pix.py:142: train_fn(Z, X, Y) takes numpy
    # arrays): every step uploads its 16 KB + 4 MB + 12 MB batch over PCIe before it is enqueued ----
    with_h2d = with_h2d_sync = None
    if not args.ablate and world == 1 and not args.minimal:
        run = (lambda: eng.enqueue_train(b)) if issue is not False else (lambda: eng.enqueue_train(b, lambda lane, e: e[1]()))
        eng.sync()
        t1 = time.perf_counter()
        for s in range(args.steps):             # strictly sequential, as the reference's loop: upload, wait, step
            eng._upload(b, Z, X, Y)
            run()
        eng.sync()
        with_h2d_sync = B * args.steps / (time.perf_counter() - t1)
        if issue is not False:
            # the product's path for host arrays (Pix2Pix.train -> GanStep.train_pipelined): the upload of batch i+1 on a
            # copy stream from page-locked staging while step i runs
            b1 = eng.built(B, 1)
            pslots = [b, b1]
            for w_ in range(6):                  # both slots record their step (call 0 eager, call 1 records)
                eng.upload_async(pslots[w_ & 1], Z, X, Y)
                eng.enqueue_train_uploaded(pslots[w_ & 1])
            eng.sync()
            eng.upload_async(pslots[0], Z, X, Y)
            t1 = time.perf_counter()
            for s in range(args.steps):
                eng.enqueue_train_uploaded(pslots[s & 1])
                eng.upload_async(pslots[(s + 1) & 1], Z, X, Y)
            eng.sync()
            with_h2d = B * args.steps / (time.perf_counter() - t1)
        else:
            with_h2d = with_h2d_sync

    ms_per_step = 1e3 * elapsed / args.steps
    value = B * world * args.steps / elapsed
    plan_gflop_per_img = nominal_flops_per_step(b) / B / 1e9
    if S == 512 and not args.config1 and B > 0:
        gflop_per_img = {'both': JOINT_GFLOP_PER_IMG, 'dcgan': DCGAN_GFLOP_PER_IMG, 'p2p': P2P_GFLOP_PER_IMG}[args.mode]
        flops_source = "SURVEY.md 8(d) constant for train_mode=%s (the loss-only forward of the other stage excluded)" % args.mode
    else:
        gflop_per_img = plan_gflop_per_img
        flops_source = "summed over the plan's convolution launches (bench.py nominal_flops_per_step)"
    out = {
        **({"INVALID": "ablation run (--ablate %s): kernels skipped, results wrong" % args.ablate} if args.ablate else {}),
        **({"INVALID": "tuning run (--repeat %s): kernels issued twice" % args.repeat} if args.repeat else {}),
        "metric": "512px heightmap+texture train images/sec", "value": round(value, 3), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("BASELINE config 1: DCGAN 64x64 generator + discriminator, " if args.config1 else "") +
                               "test1_nobn_bilin_both joint DCGAN+pix2pix train step (train_mode=%s), %dx%d, "
                               "batch %d per GPU, RMSprop lr 1e-4, LSGAN+100*L1%s"
                               % (args.mode, S, S, B, "; fp32 on v_mfma_f32_32x32x2_f32" if args.dtype == "f32" else
                                  ("; fp32 by operand splitting on the bf16 matrix cores (3 exact bf16 pieces per operand, "
                                   "6 products, fp32 accumulation)" if args.dtype == "bf16x3" else
                                   "; operands as 2 bf16 pieces (16-17 bits), 3 products on the bf16 matrix cores, fp32 "
                                   "accumulation / tensors / master weights" if args.dtype == "bf16x2" else
                                   "; products in %s on the matrix cores, fp32 accumulation / tensors / master weights"
                                   % args.dtype)),
                   "global_batch": B * world, "in_shp": S, "parallelism": "dp%d" % world,
                   **({"exchange": eng.exchange_mode} if world > 1 else {}),
                   "hip_graph": bool(args.graph), "issue": "graph" if args.graph else args.issue,
                   "host_calls_per_step": 1 if issue == 'recorded' else None,
                   "streams": len({id(d) for d in list(eng.devs) + [sd[0] for sd in eng.side if sd is not None]})},
        # ALGORITHMIC work of the step: the survey's own per-image constants for the 512x512 configs, the
        # plan-derived count (nominal_flops_per_step: same rules, summed over the plan's convolution launches) elsewhere
        "step_algorithmic_tflops": round(gflop_per_img * value / 1e3, 2),
        "step_algorithmic_gflop_per_img": round(gflop_per_img, 2), "step_flops_source": flops_source,
        "step_plan_gflop_per_img": round(plan_gflop_per_img, 2),
        "step_frac_of_peak": round(gflop_per_img * value / world / 1e3 / PEAK[args.dtype], 4),
        "step_peak_tflops": PEAK[args.dtype],
        "step_frac_of_fp32_mfma_peak": round(gflop_per_img * value / world / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)
        if args.dtype == 'f32' else None,
        # the nominal count above prices Upscale2D -> 5x5 convs at 25 MACs per output; they execute 9 (collapsed
        # 3x3 form): this is what the matrix cores actually do per second
        "step_executed_tflops": round(executed_flops_per_step / (ms_per_step * 1e-3) / 1e12 * world, 2)
        if executed_flops_per_step else None,
        "step_executed_frac_of_peak": round(executed_flops_per_step / (ms_per_step * 1e-3) / 1e12 / PEAK[args.dtype], 4)
        if executed_flops_per_step else None,
        # the reference's graph walks the DCGAN discriminator backward twice on the fake half (generator loss, discriminator
        # loss); the step walks it once and takes the generator-loss gradient as a per-sample multiple: the
        # algorithmic figure above still prices both passes, the executed one does not
        "generator_gradient_from_discriminator_pass": any(e[0] == "per_sample_ratio" for e in b.train_compute[0]),
        "losses": [float(x) for x in losses],
        # inputs uploaded from host arrays every step (the reference's train_fn(Z, X, Y) boundary); never ``value``
        "steady_state": steady,
        # the rotation: ``value`` is measured over ``resident_batches`` distinct batches; value_lr0 = the same steps with the
        # learning rate at 0 (frozen parameters); value_single_batch = ONE batch repeated (what rounds 1-5 reported)
        "resident_batches": len(pool) if rotate else 1,
        "value_lr0": round(value_lr0, 3) if value_lr0 else None,
        "value_single_batch": round(value_single, 3) if value_single else None,
        "value_with_h2d": round(with_h2d, 3) if with_h2d else None,
        "value_with_h2d_synchronous": round(with_h2d_sync, 3) if with_h2d_sync else None,
        "hbm_bound_layers": {"note": "thin first / last layers, each timed alone and warm (ten launches back to back); bound = HBM 8 TB/s; "
                                     "moved_* counts the q copy a first layer also writes",
                             "total_ms": round(sum(r["ms"] for r in thin_rows), 3), "launches": thin_rows} if thin_rows else None,
    }
    if world > 1:
        # data-parallel exchange, measured inside the timed region: the all-reduce of every (sub-)bucket on the
        # communication stream, and -- the part of it that is EXPOSED -- how long each stage stream sat in its wait
        # for the communication stream before the optimiser updates
        per = {}
        for d, i, lab in every:
            if lab is not None:
                per.setdefault(lab, []).append(every_ms[(id(d), i)])
        out["exchange"] = {
            "rccl_nranks": comm.nranks(), "bucket_mb": eng.bucket_bytes / 2 ** 20,
            "form": eng.exchange_mode,
            "collectives_per_step": len([k for k in per if k.startswith(("allreduce_", "reducescatter_", "allgather_"))]),
            "sharded_update_ms_per_step": round(sum(sum(v) / len(v) for k, v in per.items() if k.startswith(("rmsprop_shard_", "adam_shard_"))), 4),
            "allgather_ms_per_step": round(sum(sum(v) / len(v) for k, v in per.items() if k.startswith("allgather_")), 4),
            "buckets": [{"label": lab, "net": k, "MB": round(4 * n / 2 ** 20, 2),
                         "avg_ms": round(sum(per[lab]) / len(per[lab]), 4) if lab in per else None}
                        for lab, k, lo, n in b.xchg_order],
            "exposed_wait_ms_per_step": {(k[-1] if k.startswith("wait_comm") else k[len("wait_gather_"):]): round(sum(v) / len(v), 4)
                                         for k, v in per.items() if k.startswith(("wait_comm", "wait_gather_"))},
        }
    if dominant and slot:
        tot_ms = sum(timed_ms)
        avg_ms = tot_ms / slot                           # in the timed region (the other stream keeps running)
        flops_per_launch = flops_per_step / launches_per_step
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        traffic = None
        kdt_name = args.dtype if dominant.startswith(("lp_", "sp_")) else "f32"
        sfx = "f32" if kdt_name == "f32" else "bf16"
        rounds = ("r06", "r05", "r04")                                      # tools/pmc_mfma.sh; newest round first
        for pmc in ["%s_pmc_traffic_3stream_%s.json" % (r, sfx) for r in rounds] + ["r03_pmc_traffic_%s.json" % sfx,
                                                                                     "r02_pmc_traffic_%s.json" % sfx, "r01_pmc_traffic.json"]:
            pmc = os.path.join(ROOT, "profiles", pmc)
            if traffic is None and os.path.exists(pmc):
                traffic = json.load(open(pmc)).get(dominant, {}).get("hbm_bytes_per_launch")
        if traffic is None and dominant.startswith("sp_"):
            # split kernels: the PMC file names the template instantiations <KS, ST, BM, RT, WM, WN, POOL, TW, NP, ABL, CLS>; the
            # wide-tile ones of this family (pooled or not, class forms or not, as the label says), launch-weighted
            import re
            for pmc in ["%s_pmc_traffic_3stream_%s.json" % (r, args.dtype) for r in rounds]:
                pmc = os.path.join(ROOT, "profiles", pmc)
                if traffic is None and os.path.exists(pmc):
                    fam = dominant.split("<")[1].split(">")[0] + ","             # "sp_conv2_kernel<3, 1> cls" -> "3, 1,"
                    stem = dominant.split("_kernel")[0]
                    pooled, cls = "fwd+pool" in dominant, dominant.endswith(" cls")
                    rows = []
                    for k, v in json.load(open(pmc)).items():
                        if not (k.startswith(stem) and ("<" + fam) in k and v.get("hbm_bytes_per_launch")):
                            continue
                        targs = [t.strip() for t in k.split("<", 1)[1].rstrip(">").split(",")]
                        if stem == "sp_conv2":
                            if targs[7] != "32" or (targs[6] == "true") != pooled:
                                continue
                            if (len(targs) > 10 and targs[10] != "0") != cls:
                                continue
                        elif stem == "sp_wgrad":
                            if targs[4] not in ("32", "64") or (len(targs) > 6 and targs[6] != "0") != cls:
                                continue
                        rows.append(v)
                    n = sum(v["launches_sampled"] for v in rows)
                    if n:
                        traffic = sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for v in rows) / n
        iso = flops_per_launch / (isolated_ms * 1e-3) / 1e12
        # the dominant kernel is priced against the peak of the arithmetic IT runs in (a thin / small-map kernel that
        # stays fp32 in a bf16 step is an fp32 kernel)
        kdt = args.dtype if dominant.startswith(("lp_", "sp_")) else "f32"
        peak = PEAK[kdt]
        out["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "kernel_dtype": kdt,
                           "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                           "concurrent_streams": len({id(d) for d in list(eng.devs) + [sd[0] for sd in eng.side if sd is not None]}),
                           "achieved_isolated": round(iso, 2), "frac_isolated": round(iso / peak, 4),
                           "kernel": dominant, "launches_per_step": launches_per_step,
                           "avg_launch_ms": round(avg_ms, 4),
                           "algorithmic_gflop_per_launch": round(flops_per_launch / 1e9, 3),
                           "share_of_step_time": round(avg_ms * launches_per_step / ms_per_step, 3)}
    else:
        out["roofline"] = None
    if secondary_name:
        out = {"name": secondary_name, **{k: out[k] for k in (
            "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "step_algorithmic_tflops",
            "step_algorithmic_gflop_per_img", "step_flops_source", "step_frac_of_peak", "step_peak_tflops", "step_executed_tflops",
            "step_executed_frac_of_peak", "resident_batches", "losses", "value_with_h2d", "value_with_h2d_synchronous", "roofline")}}
    eng.close_pipeline()
    if comm is not None:
        comm.close()
        cdev.close()
    for d in {id(d): d for d in list(eng.devs) + [sd[0] for sd in eng.side if sd is not None]}.values():
        if d is not dev:
            d.close()
    dev.close()
    return out


LINE_LIMIT = 4096       # the driver keeps the tail of stdout: the ONE JSON line must fit (round 5's 20.8 KB line did not parse)

HEADLINE_KEYS = ("INVALID", "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_fp32_mfma", "value_lr0",
                 "value_single_batch", "resident_batches", "steady_state", "step_frac_of_peak", "step_executed_frac_of_peak",
                 "step_algorithmic_gflop_per_img", "step_peak_tflops", "generator_gradient_from_discriminator_pass", "losses",
                 "exchange", "side_file")
# dropped first when the line would still be too long (never the contract's keys)
OPTIONAL_KEYS = ("exchange", "losses", "generator_gradient_from_discriminator_pass", "step_peak_tflops",
                 "step_algorithmic_gflop_per_img", "steady_state", "resident_batches", "value_single_batch")


def headline_line(out):
    """the ONE stdout line: the contract's keys + roofline + cpu_baseline, shortened to stay below LINE_LIMIT bytes.
    Everything else (secondary configurations, the thin-layer table, notes) is in the side file."""
    line = {k: out[k] for k in HEADLINE_KEYS if k in out}
    if isinstance(line.get("config"), dict):
        c = dict(line["config"])
        if isinstance(c.get("workload"), str) and len(c["workload"]) > 260:
            c["workload"] = c["workload"][:257] + "..."
        line["config"] = c
    if isinstance(line.get("cpu_baseline"), dict):
        cb = dict(line["cpu_baseline"])
        if isinstance(cb.get("sample"), str) and len(cb["sample"]) > 200:
            cb["sample"] = cb["sample"][:197] + "..."
        if isinstance(cb.get("config1"), dict):
            cb["config1"] = {k: v for k, v in cb["config1"].items() if k != "sample"}
        line["cpu_baseline"] = cb
    if isinstance(line.get("exchange"), dict):
        line["exchange"] = {k: v for k, v in line["exchange"].items() if k != "buckets"}
    if isinstance(line.get("losses"), list):
        line["losses"] = [round(x, 6) for x in line["losses"]]
    txt = json.dumps(line)
    for k in OPTIONAL_KEYS:
        if len(txt) < LINE_LIMIT - 64:
            break
        line.pop(k, None)
        txt = json.dumps(line)
    assert len(txt) < LINE_LIMIT, len(txt)
    return txt


def main():
    args = parse_args()
    from gan_heightmaps_amd import dist
    rank, _, world = dist.env_rank_world()
    out = measure(args)
    headline = (args.mode == "both" and args.dtype == "bf16x3" and not args.config1 and args.in_shp == 512
                and args.batch_per_gpu == 4 and not args.graph and not args.one_stream and not args.no_grad_streams)
    if rank == 0 and world == 1 and headline and not args.no_secondary and not args.ablate and not args.repeat:
        import copy
        sec = []
        for name, ov in SECONDARY:
            a2 = copy.copy(args)
            a2.steps, a2.warmup, a2.profile, a2.no_cpu_baseline = args.secondary_steps, 3, False, True
            for k, v in ov.items():
                setattr(a2, k, v)
            try:
                sec.append(measure(a2, name))
            except Exception as ex:           # a secondary line must never take the headline down with it
                sec.append({"name": name, "error": "%s: %s" % (type(ex).__name__, ex)})
            r_ = sec[-1]
            # one short line per configuration on STDERR (stdout carries exactly one JSON line)
            print("secondary %-48s %s" % (name, ("%.1f img/s, %.2f ms, %s" % (r_["value"], r_["ms_per_step"], r_["dtype"]))
                                          if "value" in r_ else r_.get("error")), file=sys.stderr, flush=True)
        out["secondary"] = sec
        sp = next((r for r in sec if r.get("name") == "headline_workload_fp32_mfma" and "value" in r), None)
        if sp is not None:
            # the same workload on the fp32 matrix instruction (the headline of rounds 1-4): a named secondary value
            out["value_fp32_mfma"] = sp["value"]
            out["arithmetic_note"] = (
                "value: every fp32 operand split into three bf16 pieces that sum to it exactly, the six leading piece products "
                "on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (dtype bf16x3; peak 2500 / 6 = 416.7 TFLOP/s fp32-equivalent); "
                "passes the fp32 path's parity tests with the fp32 path's bounds in both modes (tests/test_gpu_split.py, "
                "test_gpu_step.py, test_gpu_fullsize.py).  value_fp32_mfma: the same step on v_mfma_f32_32x32x2_f32 (--dtype f32)")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.in_shp == 512 and not args.config1:
        out["cpu_baseline"] = cpu_baseline(2)
    if rank == 0:
        # the complete record (secondary configurations, thin-layer table, notes) goes to a side file ...
        try:
            side = args.side_file if os.path.isabs(args.side_file) else os.path.join(ROOT, args.side_file)
            os.makedirs(os.path.dirname(side), exist_ok=True)
            with open(side, "w") as f:
                json.dump(out, f, indent=1)
            out["side_file"] = args.side_file
        except OSError as ex:
            sys.stderr.write("bench.py: side file not written: %s\n" % ex)
        # ... and stdout carries ONE short JSON line
        print(headline_line(out), flush=True)


if __name__ == "__main__":
    main()
