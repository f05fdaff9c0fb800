"""Build libblight_mi355x.so (hand-written HIP for gfx950 + the C ABI) in-tree with hipcc.

    python -m blight_amd.build            # incremental
    python -m blight_amd.build --force

hipcc cross-compiles gfx950 without a GPU; the .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libblight_mi355x.so")
SOURCES = ["blmx_api.hip", "blmx_build.hip", "blmx_query.hip", "blmx_reads.hip", "blmx_front.hip", "blmx_stream.hip", "blmx_comm.hip", "blmx_serve.hip"]
HEADERS = ["blmx_internal.h", "device_prims.h", "index_device.h", "query_device.h", os.path.join(ROOT, "include", "blight_mi355x.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-DNDEBUG",
         "-fno-gpu-rdc"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, defines):
    """An experimental build of the same sources with extra -D flags -> scratch/lib_<name>.so (git-ignored; travels to the GPU box;
    BLIGHT_LIB_PATH selects it).  For A/B measurements only: nothing in the tree loads it."""
    out_dir = os.path.join("/tmp", "blmx_variant_" + name)
    os.makedirs(out_dir, exist_ok=True)
    os.makedirs(os.path.join(ROOT, "scratch"), exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        o = os.path.join(out_dir, src.replace(".hip", ".o"))
        objs.append(o)
        procs.append((src, subprocess.Popen([HIPCC] + FLAGS + list(defines) + ["-c", os.path.join(CSRC, src), "-o", o])))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    lib = os.path.join(ROOT, "scratch", "lib_%s.so" % name)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-lz", "-ldl", "-lpthread"])
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
