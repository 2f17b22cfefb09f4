"""Build the in-tree native library with hipcc for gfx950 (cross-compiles without a GPU)."""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_NAME = "libmidas_snps_hip.so"
LIB_PATH = os.path.join(LIB_DIR, LIB_NAME)

SOURCES = ["pack.cpp", "hostio.cpp", "row_deflate.cpp", "pack_reads.hip", "index_reads.hip", "pileup_tiles.hip", "index_direct.hip", "pileup_direct.hip", "rows_deflate.hip", "bgzf_inflate.hip", "merge_sites.hip", "genes_count.hip", "snps_abi.hip"]
HEADERS = ["layout.h", "pack.h", "kernels.h", "device_common.h", "pileup_common.h", "direct_common.h", "ctx_internal.h", "hostio.h", "row_deflate.h", "workers.h", os.path.join("..", "..", "include", "midas_snps.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build %s" % LIB_NAME)


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_native(force=False, verbose=False):
    """Compile midas_amd/csrc/* into midas_amd/lib/libmidas_snps_hip.so (gfx950 only)."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    # The atomic optimizer turns a one-lane atomicAdd into mbcnt/readfirstlane and waits for the result at once; the
    # pileup kernel fetches its next work item that way and must not stall on it (pileup_tiles.hip, dynamic items).
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
           "-amdgpu-atomic-optimizer-strategy=None", "-x", "hip"] + srcs + \
          ["-o", tmp, "-lz", "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError("hipcc failed (%d):\n%s" % (res.returncode, res.stdout))
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
