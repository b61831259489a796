#!/usr/bin/env python3
"""tests/emu/build.py - TEST INFRASTRUCTURE ONLY.

Builds tests/emu/build/libdada2hip_emu.so: the product's own sources (dada2_amd/csrc/*.hip, *.cpp, *.h) compiled for the
HOST against the functional emulator (tests/emu/hip/hip_runtime.h, emu.cpp, gcn.h).  The sources are copied into the build
directory with two mechanical rewrites the emulator needs:
  * `extern __shared__ <attrs> T name[];`  ->  `T *name = (T *)emu::dyn_lds();`   (dynamic LDS of the running block)
  * csrc/gcn.h (single-instruction wrappers in inline asm) is replaced by tests/emu/gcn.h
Nothing in dada2_amd/ knows this library exists; tests/test_emu.py opens it explicitly."""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "dada2_amd", "csrc")
OUT = os.path.join(HERE, "build")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")

DYN = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][A-Za-z0-9_ ]*?)\s+([A-Za-z_][A-Za-z0-9_]*)\[\];")


def rewrite(text):
    return DYN.sub(lambda m: f"{m.group(1)} *{m.group(2)} = ({m.group(1)} *)emu::dyn_lds();", text)


def stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    os.environ.setdefault("DADA2HIP_SYSTEM_HIP", "1")   # (the emulated library links no HIP runtime: nothing to share with torch)
    os.makedirs(os.path.join(OUT, "csrc"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "include"), exist_ok=True)
    srcs = sorted(f for f in os.listdir(SRC) if f.endswith((".hip", ".cpp", ".h")))   # (the .inc.hip files are copied too)
    deps = [os.path.join(SRC, f) for f in srcs] + [os.path.join(HERE, f) for f in ("emu.cpp", "gcn.h", "build.py", "hip/hip_runtime.h")]
    deps.append(os.path.join(ROOT, "include", "dada2hip.h"))
    lib = os.path.join(OUT, "libdada2hip_emu.so")
    if not force and not stale(lib, deps):
        return lib
    for f in srcs:
        if f == "gcn.h":
            continue
        with open(os.path.join(SRC, f)) as fh:
            text = rewrite(fh.read())
        with open(os.path.join(OUT, "csrc", f), "w") as fh:
            fh.write(text)
    shutil.copy(os.path.join(HERE, "gcn.h"), os.path.join(OUT, "csrc", "gcn.h"))
    shutil.copy(os.path.join(ROOT, "include", "dada2hip.h"), os.path.join(OUT, "include", "dada2hip.h"))
    tus = ["kernels.hip", "tail.hip", "driver.cpp", "derep.cpp", "merge.cpp", "hostsimd.cpp"]
    # one compiler process per translation unit, side by side (kernels.hip alone is two thirds of the work), then the link
    opt = os.environ.get("EMU_OPT", "-O1")
    base = [CXX, "-std=c++17", opt, "-g", "-fPIC", "-ffp-contract=off", "-w", "-I", HERE, "-x", "c++", "-c"]
    jobs = []
    objs = []
    for t in tus + ["emu.cpp"]:
        src = os.path.join(HERE, t) if t == "emu.cpp" else os.path.join(OUT, "csrc", t)
        obj = os.path.join(OUT, os.path.splitext(t)[0] + ".o")
        objs.append(obj)
        jobs.append((t, subprocess.Popen(base + [src, "-o", obj])))
    bad = [t for t, j in jobs if j.wait() != 0]
    if bad:
        raise RuntimeError("emulator build failed: " + ", ".join(bad))
    subprocess.check_call([CXX, "-shared", "-o", lib] + objs + ["-lz", "-lpthread"])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
