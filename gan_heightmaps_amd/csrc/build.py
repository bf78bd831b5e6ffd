"""Build libghm.so (gfx950 only) with hipcc.  Used by __graft_entry__.build() and by hand:
    python gan_heightmaps_amd/csrc/build.py [--force]
hipcc cross-compiles without a GPU; the .so is kept in-tree (git-ignored) so it travels with gpurun.

Every compile also records the compiler's per-kernel resource usage (VGPRs, scratch bytes, LDS) in
``<source>.resources.json`` next to the object: a kernel that starts spilling registers to scratch after an edit loses
a factor of several without failing any test, so ``check_no_spills()`` (called by build() and by tests/test_cabi.py)
fails loudly on it.
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["ctx.hip", "conv_igemm.hip", "conv_lp.hip", "conv_small.hip", "conv_thin.hip", "conv_thin_lp.hip", "conv_split.hip", "conv_pool_bwd.hip", "conv_bilinear.hip", "elementwise.hip", "elementwise_q.hip", "comm.hip"]
HEADERS = ["common.h", os.path.join("..", "..", "include", "ghm.h")]
OUT = os.path.join(os.path.dirname(HERE), "libghm.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Rpass-analysis=kernel-resource-usage"]
# GHM_BUILD_DEFINES="-DGHM_SPLIT_ABLATION ...": extra defines for tuning builds (ablation instantiations; never the shipped build)
FLAGS += os.environ.get("GHM_BUILD_DEFINES", "").split()


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def _parse_resources(text):
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]|"
                      r"VGPRs Spill|SGPRs Spill): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return out


def build(force=False, verbose=True):
    objs = []
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    procs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs) or not os.path.exists(s + ".resources.json"):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, s, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for cmd, s, p in procs:
        _, err = p.communicate()
        other = [l for l in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and l.strip()]
        if other and verbose:
            print("\n".join(other), file=sys.stderr)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        with open(s + ".resources.json", "w") as f:
            json.dump(_parse_resources(err), f, indent=0, sort_keys=True)
    if force or procs or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    check_no_spills()
    return OUT


def kernel_resources():
    out = {}
    for src in SOURCES:
        p = os.path.join(HERE, src + ".resources.json")
        if os.path.exists(p):
            out.update(json.load(open(p)))
    return out


# scratch use that is known and outside every hot loop (bytes per lane): the 36-tap fan-out variant spills in its
# prologue; the pooled fan-out variants, which hold two rows of accumulators, reload 11 spilled weight fragments per
# 104-MFMA pixel group; the 128-filter data-gradient class form (conv_split.hip CLS = 2) keeps 32 bytes outside its slab loop
BENIGN_SCRATCH = {"sp_conv2_kernelILi3ELi1ELi128ELi8ELi1ELi4E": 24, "sp_conv2_kernelILi3ELi1ELi128ELi8ELi2ELi4E": 24,
                  "sp_conv2_kernelILi3ELi1ELi128ELi8ELi2ELi4ELb0ELi32ELi3ELi0ELi2E": 32, "sp_conv2_kernelILi3ELi1ELi128ELi8ELi2ELi4ELb0ELi32ELi2ELi0ELi2E": 32, "fanout_kernelILi18ELi2ELi4ELb0ELb0": 52, "fanout_kernelILi13ELi2ELi2ELb0ELb1": 96,
                  "fanout_kernelILi18ELi2ELi2ELb0ELb1": 160, "sp_wgrad_pooled_kernelILi64ELi3E": 36}


def check_no_spills():
    """no kernel of the library may use scratch memory (register spills / dynamically indexed private arrays) beyond
    the allow-list above"""
    def allowed(name, v):
        return any(tag in name and v.get("ScratchSize", 0) <= cap for tag, cap in BENIGN_SCRATCH.items())
    bad = {k: v for k, v in kernel_resources().items()
           if (v.get("ScratchSize", 0) > 0 or v.get("VGPRs Spill", 0) > 0) and not allowed(k, v)}
    if bad:
        raise RuntimeError("kernels using scratch memory (spills): " +
                           ", ".join("%s: %d B/lane" % (k, v.get("ScratchSize", 0)) for k, v in sorted(bad.items())))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
