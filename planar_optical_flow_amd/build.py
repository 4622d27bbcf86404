"""Build libpof_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m planar_optical_flow_amd.build [--force]

Every csrc/*.hip is compiled to an object (in parallel) and linked into
planar_optical_flow_amd/lib/libpof_hip.so.  hipcc cross-compiles without a GPU.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpof_hip.so")
OBJDIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
    "-ffp-contract=off",  # float64 index math must round like NumPy; FMAs are explicit
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
    "-I", os.path.join(REPO, "include"), "-I", CSRC,
]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return _sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(REPO, "include", "pof_abi.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    hdr_t = max(os.path.getmtime(p) for p in _deps() if p.endswith(".h"))
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
        return obj
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for o in glob.glob(os.path.join(OBJDIR, "*.o")):
            os.remove(o)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
