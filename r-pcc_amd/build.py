"""Builds librpcc_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "rpcc_hip.hip")
DEPS = [os.path.join(_HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(_HERE, "csrc")))] + \
       [os.path.join(os.path.dirname(_HERE), "include", "rpcc_hip.h")]
LIB = os.path.join(_HERE, "lib", "librpcc_hip.so")
HOST_SRC = os.path.join(_HERE, "csrc", "rpcc_host.c")
HOST_LIB = os.path.join(_HERE, "lib", "librpcc_host.so")

# -ffp-contract=off: the reference's C++ (projection, models, prediction, quantisation) is un-fused x86 SSE arithmetic and a
# contracted FMA changes results.  (The reference's CUDA FPS kernel is a different matter: nvcc contracts its distance into
# FMAs by default; the build's FPS follows its own un-fused specification, see DESIGN.md section 2 "parity unpinned".)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
               "-shared", "-Wno-unused-value"]


def source_digest():
    """sha256 over the sources librpcc_hip.so is built from (csrc/*, include/rpcc_hip.h; names and bytes, sorted): what a committed set of
    profiler counters (profiles/pmc_current*.json) is tied to -- bench.py prints no counter-derived fraction for another build."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read())
        h.update(b"\0")
    return h.hexdigest()


def build_host(force=False, verbose=False):
    """librpcc_host.so: the plain-C container packer (bzip2 through the libbz2 the interpreter's bz2 module links)."""
    os.makedirs(os.path.dirname(HOST_LIB), exist_ok=True)
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= os.path.getmtime(HOST_SRC):
        return HOST_LIB
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-Wall", HOST_SRC, "-o", HOST_LIB, "-l:libbz2.so.1.0"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_LIB


def build(force=False, verbose=False):
    try:
        build_host(force, verbose)
    except (subprocess.CalledProcessError, OSError) as e:   # no libbz2.so.1.0 / no gcc: compress_utils.pack_frames then
        print("librpcc_host.so not built (%s): containers are packed by the interpreter's bz2 module" % e)  # takes the Python path
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("RPCC_EXTRA_FLAGS", "").split()   # developer builds only: -DRPCC_DEVTRACE (csrc/rpcc_trace.h)
    cmd = [hipcc] + HIPCC_FLAGS + extra + [SRC, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
