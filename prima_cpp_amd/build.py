"""Builds libprima_mi355.so (hand-written HIP for gfx950) in-tree with hipcc. No JIT, no torch extension
machinery: the .so travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libprima_mi355.so")
SOURCES = ["c_api.hip", "quantize.hip", "mmvq.hip", "mmvq_cols.hip", "repack.hip", "layer_ops.hip", "ggml_ops.hip", "engine.hip", "mmq.hip", "mmq_i8.hip", "probe.hip", "engine_probe.hip", "attn_prefill.hip", "attn_split.hip", "attn_flash.hip", "attn_q8.hip", "attn_wo.hip", "ring.hip", "upload.hip"]
# -ffp-contract=off: the quantizers must round exactly like the reference (no fused multiply-add where
# the reference has a separate multiply and add); FMAs we want are written as fmaf().
EXTRA = os.environ.get("PM355_EXTRA_FLAGS", "").split()
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-const-variable", "-Wno-unused-value", "-Wno-unused-function", "-Wno-unused-result"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(os.path.dirname(HERE), "include", "prima_mi355.h")]
    objs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    if force or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
