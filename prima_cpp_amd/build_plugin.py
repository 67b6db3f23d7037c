"""Builds libggml-mi355.so, the ggml-backend plug-in, against the HOST PROJECT's ggml headers (the reference under
/root/reference in this container; nothing is copied). Plain g++: the plug-in contains no device code, it calls the C ABI
of libprima_mi355.so. When the reference tree is absent (GPU box) the prebuilt .so that travelled with the repo is kept."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PRIMA_REFERENCE", "/root/reference")
SRC = os.path.join(HERE, "csrc", "ggml_backend_mi355.cpp")
OUT = os.path.join(HERE, "libggml-mi355.so")


def build(force=False):
    inc = [os.path.join(REF, "ggml", "include"), os.path.join(REF, "ggml", "src")]
    if not os.path.exists(os.path.join(inc[0], "ggml-backend.h")):
        print(f"build_plugin: {REF} not present - keeping prebuilt {OUT}" if os.path.exists(OUT) else
              f"build_plugin: {REF} not present and no prebuilt plug-in: skipped")
        return OUT if os.path.exists(OUT) else None
    deps = [SRC, os.path.join(HERE, "csrc", "ggml_graph_plan.h"), os.path.join(HERE, "..", "include", "prima_mi355.h"), os.path.join(HERE, "..", "include", "ggml_backend_mi355.h"),
            os.path.join(HERE, "libprima_mi355.so")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps if os.path.exists(d)):
        return OUT
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-Wno-missing-field-initializers"] + [f"-I{i}" for i in inc] + [SRC, "-o", OUT, f"-L{HERE}", "-lprima_mi355",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
