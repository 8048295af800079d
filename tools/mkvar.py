"""Development helper: build a variant of libsemicrf_hip.so in which ONLY persist.hip (or the files named with --files) is
recompiled with extra -D defines; every other object comes from the release build's csrc/_obj.
   python tools/_mkvar.py NAME DEFINE[=V] ... [--files a.hip,b.hip]"""
import os, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transkun_amd import _build

def main():
    args = sys.argv[1:]
    files = ["persist.hip"]
    if "--files" in args:
        i = args.index("--files"); files = args[i + 1].split(","); del args[i:i + 2]
    name, defines = args[0], args[1:]
    _build.build()
    objdir = os.path.join(_build.CSRC, "_obj_" + name); os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + _build.ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-Wno-unused-function"]
    objs, procs = [], []
    for src in _build.sources():
        base = os.path.basename(src)
        if base in files:
            obj = os.path.join(objdir, base[:-4] + ".o")
            cmd = [_build._hipcc()] + flags + ["-D" + d for d in defines] + _build.EXTRA_FLAGS.get(base, []) + ["-c", src, "-o", obj]
            if "--save-temps" in os.environ.get("MKVAR_EXTRA", ""):
                cmd.insert(1, "-save-temps=obj")
            procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        else:
            obj = os.path.join(_build.CSRC, "_obj", base[:-4] + ".o")
        objs.append(obj)
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode()); raise SystemExit(1)
    vdir = os.path.join(_build.HERE, "_variants", name); os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, "libsemicrf_hip.so")
    subprocess.check_call([_build._hipcc(), "--offload-arch=" + _build.ARCH, "-shared", "-fPIC", "-o", out] + objs)
    shutil.copy(_build.build_torch_shim(), os.path.join(vdir, "libsemicrf_torch.so"))
    print(out)

main()
