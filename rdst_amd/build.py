"""In-tree build of the HIP extension (gfx950 only) — explicit hipcc, no JIT cache, so the
built ``librdst_hip.so`` travels with the repo snapshot to the GPU box."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = [os.path.join(HERE, "csrc", "rdst_kernels.hip"), os.path.join(HERE, "csrc", "rdst_tuner.cpp"),
           os.path.join(HERE, "csrc", "rdst_regions.cpp")]
HEADERS = [os.path.join(ROOT, "include", "rdst_hip.h")]
OUT = os.path.join(HERE, "librdst_hip.so")


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the device route cannot be built (there is no CPU fallback)")


STAMP = OUT + ".srchash"   # what the library was built from (content, not mtimes: a checkout or a snapshot copy changes those)


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for p in SOURCES + HEADERS:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), *SOURCES, "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(_source_hash() + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
