"""Developer script (GPU box): how fast host threads read the library's page-locked memory (midas_snps_host_alloc) compared
with ordinary memory -- the row formatter reads its input from such buffers.  usage: python tools/pinned_read_rate.py"""
import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, '.')
from midas_amd import abi
lib = abi.load_library()
ctx = abi.Context(0)
n = 64 << 20
p = lib.midas_snps_host_alloc(n)
a = np.frombuffer((C.c_uint8 * n).from_address(p), dtype=np.uint32)
b = np.empty_like(a)
b[:] = 1
a[:] = 1
for name, x in (("pageable", b), ("pinned", a), ("pageable", b), ("pinned", a)):
    t = time.perf_counter(); s = int(x.sum()); dt = time.perf_counter() - t
    print("%s read: %.1f ms for 64 MiB -> %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9))
t = time.perf_counter(); b[:] = a; dt = time.perf_counter() - t
print("copy pinned -> pageable: %.1f ms" % (dt * 1e3))
t = time.perf_counter(); a[:] = b; dt = time.perf_counter() - t
print("copy pageable -> pinned: %.1f ms" % (dt * 1e3))
