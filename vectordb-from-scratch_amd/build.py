"""Build libvdbflat.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvdbflat.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs += [os.path.join(HERE, "..", "include", h) for h in ("vdb_flat.h", "vdb_hnsw.h", "vdb_shard.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    if force or _stale():
        out = None if verbose else subprocess.DEVNULL
        subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=out, stderr=None if verbose else subprocess.STDOUT)
    return LIB


if __name__ == "__main__":
    print(build(force=False, verbose=True))
