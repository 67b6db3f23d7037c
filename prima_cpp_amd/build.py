"""Builds libprima_mi355.so (hand-written HIP for gfx950) in-tree with hipcc. No JIT, no torch extension
machinery: the .so travels with the repo snapshot to the GPU box.

build()            the product library prima_cpp_amd/libprima_mi355.so
build_experiments()  libprima_mi355_exp.so = the product sources + decode_engine.hip with -DPM_EXPERIMENTS=1: round 5's three measured-slower forms of the decode
                   layer live only there (tests/test_gpu_experiments.py runs their tests against it)
build_probe()      tools/csrc -> prima_cpp_amd/libprima_mi355_probe.so: measurement helpers (HBM streaming-read ceiling, persistent-layer
                   skeleton) that bench.py / tools/ load separately - nothing of them is inside the product library
build(tag=, extra=)  an A/B or measurement variant of the product library with extra compiler flags -> ab/<tag>.so, objects under
                   ab/obj_<tag>/ (select it at run time with PM355_LIB=ab/<tag>.so)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libprima_mi355.so")
PROBE_LIB = os.path.join(HERE, "libprima_mi355_probe.so")
SOURCES = ["c_api.hip", "quantize.hip", "mmvq.hip", "mmvq_cols.hip", "repack.hip", "layer_ops.hip", "ggml_ops.hip", "engine.hip", "mmq.hip", "mmq_pf.hip", "mmq_i8.hip", "mmq_big.hip",
           "attn_prefill.hip", "attn_cached.hip", "attn_flash_mfma.hip", "attn_split.hip", "attn_flash.hip", "attn_q8.hip", "ring.hip", "upload.hip", "ts.hip"]
# round 5's measured-slower forms of the decode layer (sum-of-squares partials, attention tail, persistent engine): a library of their own, loaded by their tests
EXP_LIB = os.path.join(HERE, "libprima_mi355_exp.so")
EXP_SOURCES = SOURCES + ["decode_engine.hip"]
PROBE_SOURCES = ["probe.hip", "engine_probe.hip", "probe_api.hip", "overlap_probe.hip"]
# -ffp-contract=off: the quantizers must round exactly like the reference (no fused multiply-add where
# the reference has a separate multiply and add); FMAs we want are written as fmaf().
EXTRA = os.environ.get("PM355_EXTRA_FLAGS", "").split()
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs with the dispatch instead of through an s_load of the kernarg segment
# (mmvq.hip hands the six values its prologue starts with that way: +0.4 % on the 70B decode step, +0.9 % on the 8B one, interleaved A/B on one box).
# The option is an internal LLVM one: it goes to mmvq.hip only, and only when this compiler accepts it (probed once; a ROCm / LLVM upgrade that renames it
# builds without it instead of failing everything).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-const-variable", "-Wno-unused-value", "-Wno-unused-function", "-Wno-unused-result"]
KERNARG_PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
# (round 5: first restricted to mmvq.hip - the 2..64-token steps, ten small launches per layer, then ran 1.3-2.4 % slower than round 4's: every kernel of the
#  product library gets it again, behind the probe; the probe library does not)
PER_SOURCE_FLAGS = {s_: KERNARG_PRELOAD for s_ in SOURCES}
_probe_ok = {}


def _flags_accepted(hipcc, flags):
    key = (hipcc, tuple(flags))
    if key not in _probe_ok:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "p.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void k(int * p, int v) { p[0] = v; }\n")
            r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-c", src, "-o", os.path.join(d, "p.o")] + list(flags),
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            _probe_ok[key] = r.returncode == 0
    return _probe_ok[key]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(srcs, hdrs, objdir, lib, flags, force, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            extra = PER_SOURCE_FLAGS.get(os.path.basename(s), [])
            if extra and not _flags_accepted(hipcc, extra):
                extra = []
            cmd = [hipcc] + flags + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
            if len(procs) >= (os.cpu_count() or 4):
                c, pr = procs.pop(0)
                if pr.wait():
                    raise subprocess.CalledProcessError(pr.returncode, c)
    for c, pr in procs:
        if pr.wait():
            raise subprocess.CalledProcessError(pr.returncode, c)
    if force or _newer(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "prima_mi355.h")]


def build(force=False, verbose=False, tag=None, extra=()):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if tag:
        ab = os.path.join(ROOT, "ab")
        return _compile(srcs, _headers(), os.path.join(ab, "obj_" + tag), os.path.join(ab, tag + ".so"), EXTRA + list(extra) + FLAGS, force, verbose)
    return _compile(srcs, _headers(), CSRC, LIB, EXTRA + FLAGS, force, verbose)


def build_experiments(force=False, verbose=False):
    """prima_cpp_amd/libprima_mi355_exp.so: the product sources + decode_engine.hip with -DPM_EXPERIMENTS=1 (objects under csrc/obj_exp/)"""
    srcs = [os.path.join(CSRC, s) for s in EXP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    for s_ in EXP_SOURCES:
        PER_SOURCE_FLAGS.setdefault(s_, KERNARG_PRELOAD)
    return _compile(srcs, _headers(), os.path.join(CSRC, "obj_exp"), EXP_LIB, EXTRA + ["-DPM_EXPERIMENTS=1"] + FLAGS, force, verbose)


def build_probe(force=False, verbose=False):
    d = os.path.join(ROOT, "tools", "csrc")
    srcs = [os.path.join(d, s) for s in PROBE_SOURCES]
    hdrs = _headers() + [os.path.join(d, "pm355_probe.h")]
    return _compile(srcs, hdrs, os.path.join(d, "obj"), PROBE_LIB, FLAGS + ["-I" + CSRC], force, verbose)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if args:                                   # python -m prima_cpp_amd.build <tag> [-DFLAG ...]
        print(build(force="--force" in sys.argv, verbose=True, tag=args[0], extra=[a for a in sys.argv[2:] if a.startswith("-D")]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
        print(build_experiments(force="--force" in sys.argv, verbose=True))
        print(build_probe(force="--force" in sys.argv, verbose=True))
