"""Build the gfx950 HIP library in-tree with hipcc (no hipify, no torch cpp_extension).

`python -m transkun_amd._build` or transkun_amd._build.build().  The output
transkun_amd/libsemicrf_hip.so is git-ignored but travels to the GPU box with the snapshot.
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsemicrf_hip.so")
ARCH = "gfx950"
# persist.hip never produces or consumes NaNs in its math; without this clang canonicalises every fmaxf
# operand with an extra v_max (the dependent chain of the spine is instruction-count bound).
EXTRA_FLAGS = {"persist.hip": ["-fno-honor-nans"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, defines, verbose: bool = False) -> str:
    """Development helper: build libsemicrf_<name>.so with extra -D defines (timing ablations)."""
    objdir = os.path.join(HERE, "csrc", "_obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
             "-Wno-unused-function"] + ["-D" + d for d in defines]
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [_hipcc()] + flags + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed")
    vdir = os.path.join(HERE, "_variants", name)
    os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, "libsemicrf_hip.so")
    subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs)
    # the torch-ops shim next to it (it finds "libsemicrf_hip.so" by $ORIGIN): SEMICRF_LIB=<vdir>/libsemicrf_hip.so
    shutil.copy(build_torch_shim(), os.path.join(vdir, "libsemicrf_torch.so"))
    return out


MARSHAL = os.path.join(HERE, "_semicrf_marshal.so")


def build_marshal(force: bool = False) -> str:
    """The CPython helper that packs / unpacks interval lists (host code, gcc)."""
    import sysconfig
    src = os.path.join(CSRC, "pymarshal.c")
    if not force and os.path.exists(MARSHAL) and os.path.getmtime(MARSHAL) > os.path.getmtime(src):
        return MARSHAL
    cc = os.environ.get("CC") or shutil.which("gcc") or "cc"
    tmp = MARSHAL + ".tmp"
    subprocess.check_call([cc, "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], src, "-o", tmp])
    os.replace(tmp, MARSHAL)
    return MARSHAL


TORCH_LIB = os.path.join(HERE, "libsemicrf_torch.so")


def build_torch_shim(force: bool = False, verbose: bool = False) -> str:
    """libsemicrf_torch.so: the LibTorch stable-ABI registration of the C ABI as torch.ops.semicrf.* (host C++ only:
    torch/csrc/stable headers + the aoti C shim; links libsemicrf_hip.so by $ORIGIN and the torch libraries the process
    has loaded anyway).  No hipify, no torch.utils.cpp_extension."""
    import torch
    src = os.path.join(CSRC, "torch_ops.cpp")
    cpu_src = os.path.join(CSRC, "cpu_ops.cpp")           # the CPU dispatch key's host kernels (plain C++, OpenMP)
    deps = [src, cpu_src, os.path.join(CSRC, "cpu_ops.h"), os.path.join(HERE, "..", "include", "semicrf_hip.h"), LIB]
    if not force and os.path.exists(TORCH_LIB) and os.path.getmtime(TORCH_LIB) > max(os.path.getmtime(d) for d in deps):
        return TORCH_LIB
    troot = os.path.dirname(torch.__file__)
    cxx = os.environ.get("CXX") or shutil.which("g++") or "c++"
    tmp = TORCH_LIB + ".tmp"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-fno-fast-math", "-I" + os.path.join(troot, "include"), src, cpu_src,
           "-o", tmp,
           "-L" + HERE, "-l:libsemicrf_hip.so", "-L" + os.path.join(troot, "lib"), "-ltorch_cpu", "-ltorch_hip", "-lc10",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(troot, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, TORCH_LIB)
    return TORCH_LIB


DEBUG_LIB = os.path.join(HERE, "libsemicrf_hip_debug.so")


def build_debug(force: bool = False, verbose: bool = False) -> str:
    """libsemicrf_hip_debug.so: every source compiled with -DSEMICRF_DEBUG_BUILD=1 -- the reference tile kernels of the interval
    scorer (scorer_mfma.hip: interval_score_tile_kernel<64|128>), all of semicrf_debug_score_variant's choices and the launch
    geometry knobs of tools/ (read from the environment).  The parity tests load it explicitly (through ctypes, the same C ABI)
    as the bit-level reference of the default kernels; nothing under transkun_amd/ uses it."""
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    if not force and os.path.exists(DEBUG_LIB) and os.path.getmtime(DEBUG_LIB) > max(os.path.getmtime(d) for d in deps):
        return DEBUG_LIB
    objdir = os.path.join(CSRC, "_obj_debug")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-Wno-unused-function",
             "-DSEMICRF_DEBUG_BUILD=1"]
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [_hipcc()] + flags + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed (debug library)")
    tmp = DEBUG_LIB + ".tmp"
    subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs)
    os.replace(tmp, DEBUG_LIB)
    return DEBUG_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    build_marshal(force)
    path = _build_hip(force, verbose)
    build_torch_shim(force, verbose)
    build_debug(force, verbose)
    return path


def _build_hip(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    objdir = os.path.join(HERE, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
             "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
    procs = []
    objs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(x) for x in [src] + hdrs)):
            continue
        extra = EXTRA_FLAGS.get(os.path.basename(src), [])
        cmd = [_hipcc()] + flags + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out):
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            failed = True
    if failed:
        raise RuntimeError("hipcc failed")
    tmp = LIB + ".tmp"
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
