"""Build ``libgnan_hip.so`` in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every ``csrc/*.hip`` file becomes its own object under ``build/`` (compiled in parallel, re-compiled only when it or a
header changed); the objects are linked into the one shared library the C ABI lives in.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "libgnan_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def headers():
    return glob.glob(os.path.join(HERE, "csrc", "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def _stale(target: str, deps) -> bool:
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def needs_build() -> bool:
    return _stale(OUT, sources() + headers())


def _compile(src: str, force: bool, verbose: bool) -> str:
    obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
    if force or _stale(obj, [src] + headers()):
        cmd = ["hipcc", *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return obj


def build(force: bool = False, verbose: bool = True, out: str = OUT) -> str:
    if not force and out == OUT and not needs_build():
        _stamp(out)
        return out
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, verbose), sources()))
    # exports: exactly the C ABI (include/gnan_hip.h).  -fvisibility=hidden hides the library's own C++ helpers; the version
    # script also keeps template instantiations of the standard library (default visibility by their namespace) local
    vs = os.path.join(OBJ_DIR, "exports.map")
    with open(vs, "w") as f:
        f.write("{ global: gnan_*; local: *; };\n")
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vs, *objs, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    _stamp(out)
    return out


def _stamp(out: str) -> None:
    """Record the commit the library was built from next to it (bench.py reports it where there is no .git)."""
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                             timeout=5).stdout.strip()
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True,
                               text=True, timeout=10).stdout.strip()
        if sha:
            with open(os.path.splitext(out)[0] + ".sha", "w") as f:
                f.write(sha + ("+dirty" if dirty else "") + "\n")
    except Exception:
        pass


if __name__ == "__main__":
    build(force="--force" in sys.argv)
