"""Developer script (GPU box): does a large allocation stall less when it is made in pieces?  (tools/alloc_probe.py: a 58 GB
hipMalloc waits seconds when it lands on memory a process before it released and the driver is still wiping.)
Rounds of: a process that allocates GB in one piece, WRITES it and exits; at once a process that allocates GB again -- in ONE piece,
or in pieces of PIECE_GB -- and reports how long the allocation took.   usage: python tools/alloc_probe2.py [GB=58] [PIECE_GB=2] [rounds=8]"""
import subprocess
import sys
import time

CHILD = r'''
import ctypes as C, sys, time
hip = C.CDLL("libamdhip64.so")
gb, piece, write = float(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
hip.hipSetDevice(0); hip.hipFree(None)
n, m = int(gb * 1e9), int(piece * 1e9)
ps, worst = [], 0.0
t0 = time.perf_counter()
left = n
while left > 0:
    p = C.c_void_p()
    t = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(min(m, left)))
    worst = max(worst, time.perf_counter() - t)
    assert rc == 0, rc
    ps.append((p, min(m, left)))
    left -= m
total = time.perf_counter() - t0
if write:
    for p, k in ps:
        hip.hipMemset(p, 0xA5, C.c_size_t(k))
    hip.hipDeviceSynchronize()
print("%.4f %.4f %d" % (total, worst, len(ps)))
'''


def child(gb, piece, write):
    r = subprocess.run([sys.executable, "-c", CHILD, str(gb), str(piece), str(int(write))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    return r.stdout.strip() or r.stderr[-200:]


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 58
    piece = float(sys.argv[2]) if len(sys.argv) > 2 else 2
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    for how, pc in (("ONE piece", gb), ("pieces of %g GB" % piece, piece), ("ONE piece", gb), ("pieces of %g GB" % piece, piece)):
        out = []
        for _ in range(rounds):
            child(gb, gb, True)                 # the process before: everything written, released at exit
            out.append(child(gb, pc, True))     # (writes too: it is the next round's "process before" as well)
        print("%-18s after a writer's exit, %d rounds: total s / worst single hipMalloc s / pieces: %s" % (how, rounds, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
