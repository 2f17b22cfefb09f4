"""Build the in-tree native library with hipcc for gfx950 (cross-compiles without a GPU).

Every source is compiled to its own object (in parallel, kept under midas_amd/lib/obj/ and reused while neither the source
nor any header changed), then the objects are linked into midas_amd/lib/libmidas_snps_hip.so.
"""

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_NAME = "libmidas_snps_hip.so"
LIB_PATH = os.path.join(LIB_DIR, LIB_NAME)

SOURCES = ["contigs.cpp", "hostio.cpp", "row_deflate.cpp", "comm.cpp", "pack_reads.hip", "index_reads.hip", "pileup_tiles.hip", "index_direct.hip", "pileup_direct.hip", "pileup_long.hip", "rows_deflate.hip", "bgzf_inflate.hip", "bam_walk.hip", "measure.hip", "merge_sites.hip", "genes_count.hip", "device_sort.hip", "snps_abi.hip"]
HEADERS = ["layout.h", "contigs.h", "kernels.h", "device_common.h", "pileup_common.h", "direct_common.h", "ctx_internal.h", "hostio.h", "row_deflate.h", "workers.h", "crc32.h", os.path.join("..", "..", "include", "midas_snps.h")]

# The atomic optimizer turns a one-lane atomicAdd into mbcnt/readfirstlane and waits for the result at once; the
# pileup kernel fetches its next work item that way and must not stall on it (pileup_tiles.hip, dynamic items).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build %s" % LIB_NAME)


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _newest_header():
    t = 0.0
    for h in HEADERS:
        p = os.path.join(CSRC, h)
        if os.path.exists(p):
            t = max(t, os.path.getmtime(p))
    return max(t, os.path.getmtime(os.path.abspath(__file__)))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    if _newest_header() > t:
        return True
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in _sources())


def _compile(obj, cmd, verbose):
    """One object, written under a per-process name and renamed into place: ranks of one launch that all find the library
    missing may compile the same source at once and none of them ever links a half-written object."""
    tmp = obj + ".tmp.%d" % os.getpid()
    try:
        _run(cmd + [tmp], verbose)
        os.replace(tmp, obj)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed (%d): %s\n%s" % (res.returncode, " ".join(cmd), res.stdout))


def build_native(force=False, verbose=False, extra_flags=(), lib_path=None, replace=None):
    """Compile midas_amd/csrc/* into midas_amd/lib/libmidas_snps_hip.so (gfx950 only).

    `extra_flags` / `lib_path` / `replace`: developer variants (tools/build_variant.sh): other -D switches, another output
    name, another file in place of a source ({"pileup_direct.hip": "tools/variants/x.hip"}); their objects are not cached.
    """
    variant = bool(extra_flags) or lib_path is not None or bool(replace)
    out = lib_path or LIB_PATH
    if not force and not variant and not _stale():
        return out
    hipcc = _hipcc()
    obj_dir = OBJ_DIR if not variant else OBJ_DIR + ".variant.%d" % os.getpid()
    os.makedirs(obj_dir, exist_ok=True)
    hdr_t = _newest_header()
    jobs = []
    objs = []
    for s in _sources():
        src = os.path.abspath(replace[s]) if replace and s in replace else os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".", "_") + ".o")
        objs.append(obj)
        if force or variant or not os.path.exists(obj) or os.path.getmtime(obj) < max(hdr_t, os.path.getmtime(src)):
            jobs.append((obj, [hipcc] + FLAGS + list(extra_flags) + ["-I", CSRC, "-x", "hip", "-c", src, "-o"]))
    try:
        workers = max(1, min(len(jobs), (os.cpu_count() or 4)))
        if jobs:
            with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
                for f in [pool.submit(_compile, o, j, verbose) for (o, j) in jobs]:
                    f.result()
        tmp = out + ".tmp.%d" % os.getpid()
        try:
            _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp, "-lz", "-lpthread", "-ldl"], verbose)
            os.replace(tmp, out)
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
    finally:
        if variant:
            shutil.rmtree(obj_dir, ignore_errors=True)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    force = "--force" in args
    verbose = "-v" in args
    out = None
    extra = []
    repl = {}
    it = iter(a for a in args if a not in ("--force", "-v"))
    for a in it:
        if a == "-o":
            out = next(it)
        elif a == "--replace":
            k, v = next(it).split("=", 1)
            repl[k] = v
        else:
            extra.append(a)
    print(build_native(force=force, verbose=verbose, extra_flags=extra, lib_path=out, replace=repl))
