"""Developer script (GPU box): touch the device's whole memory once, in a process of its own.

Two things a hipMalloc can wait for on these boxes (profiles/r06_alloc_probe.txt, r06_alloc_probe2.txt, r06_cli_stage_c4*.txt):
  (A) VRAM that nobody has used since the device was reset is cleared when it is FIRST allocated, at ~15 GB/s: on a freshly
      leased box a 58 GB hipMalloc takes 2.5 - 4.2 s, once per region -- after every region has been used and released once the
      same allocation takes 0.3 ms, in one piece or in pieces;
  (B) what a process releases is wiped by the driver behind its back, at ~100 GB/s: a process that starts within half a second of
      the exit of one that held 55 GB may wait up to that long at one of its allocations (and right behind THIS prelude's exit --
      300 GB released at once -- a context's creation took 1.5 s: hence the pause at the end).
(A) is the box's first use, not the product: the timing tools (c4_files.py, cli_stage.py, predict_ranks.py) run this first.
(B) is what back-to-back samples really see; c4_files.py's C4_PAUSE separates the runs when the stage alone is wanted.
usage: python tools/vram_prelude.py [fraction of the free memory = 0.97]"""
import subprocess
import sys
import time

CHILD = r'''
import ctypes as C, sys
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0); hip.hipFree(None)
free, total = C.c_size_t(), C.c_size_t()
hip.hipMemGetInfo(C.byref(free), C.byref(total))
want, piece, got = int(free.value * float(sys.argv[1])), 2 << 30, 0
ps = []
while got + piece <= want:
    p = C.c_void_p()
    if hip.hipMalloc(C.byref(p), C.c_size_t(piece)) != 0:
        break
    hip.hipMemset(p, 0, C.c_size_t(piece))
    ps.append(p); got += piece
hip.hipDeviceSynchronize()
print("%.1f of %.1f GB" % (got / 1e9, total.value / 1e9))
'''


def run(fraction=0.97, quiet=False):
    t = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", CHILD, str(fraction)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if not quiet:
        print("vram prelude: touched %s in %.1f s (a fresh box's first use of its memory, taken out of what follows)" % (
            r.stdout.strip() or r.stderr[-200:], time.perf_counter() - t), flush=True)
    time.sleep(4.0)       # (the driver wipes what this process released: ~300 GB)


if __name__ == "__main__":
    run(float(sys.argv[1]) if len(sys.argv) > 1 else 0.97)
