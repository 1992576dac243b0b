"""Builds pathtracer_amd/libptcore.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libptcore.so")
SOURCES = ["ptcore.hip", "bvh_build.cpp", "ply.cpp", "scene_prep.cpp", "frame_dump.cpp"]
HEADERS = ["device_math.h", "trace_kernels.h", "frame_kernels.h", "sort_kernels.h", "../../include/ptcore.h",
           "context_impl.h", "launch_impl.h", "pipeline_impl.h", "train_impl.h", "batch_impl.h", "scene_impl.h", "comm_impl.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the reference's arithmetic is restated operation by operation; a fused
# multiply-add would change results.  Division and sqrt stay correctly rounded (hipcc default).
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", "-fvisibility=hidden",
         "-pthread", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=True):
    if not force and not needs_build():
        ex = os.path.join(os.path.dirname(HERE), "examples")
        exe, src = os.path.join(ex, "pt_host"), os.path.join(ex, "pt_host.cpp")
        if not os.path.exists(exe) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(exe)):
            build_example(verbose)
        return LIB
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", path, "-o", path + ".o"]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "c++")
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(path + ".o")
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-ldl", "-lz", "-pthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    build_example(verbose)
    return LIB


def build_example(verbose=True):
    """examples/pt_host: the C++ host above the C ABI (host code only; links the library just built)."""
    ex = os.path.join(os.path.dirname(HERE), "examples")
    src = os.path.join(ex, "pt_host.cpp")
    if not os.path.exists(src):
        return
    cmd = [HIPCC, "-x", "c++", "-O2", "-std=c++17", "-Wall", src, "-o", os.path.join(ex, "pt_host"), "-L" + HERE, "-lptcore",
           "-Wl,-rpath,$ORIGIN/../pathtracer_amd"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
