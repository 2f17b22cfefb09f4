"""Developer script (GPU box): what a large hipMalloc costs, and when.

The pileup stage's device decode takes ONE arena (~2.3 x the BAM's inflated bytes: 58 GB for BASELINE.json configs[3] through the
files).  Its hipMalloc was timed at 3.5 ms on some runs and at 0.5 - 2.3 s on others.  This probe separates the candidates:
  (a) memory another process has just written and released is cleared before it is handed out again (cost ~ bytes, paid by whoever
      allocates next; waiting a few seconds first does not help), or
  (b) the release itself wipes (asynchronously) and the next allocation waits for that (waiting helps).
Each step runs in a process of its own (the CLI is one process per sample).
usage: python tools/alloc_probe.py [GB=58]"""
import ctypes as C
import subprocess
import sys
import time

CHILD = r'''
import ctypes as C, sys, time
hip = C.CDLL("libamdhip64.so")
gb, dirty, free = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(gb * 1e9)
t0 = time.perf_counter(); hip.hipSetDevice(0); hip.hipFree(None); t_ctx = time.perf_counter() - t0
p = C.c_void_p()
t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(n)); t_malloc = time.perf_counter() - t0
t_set = 0.0
if dirty:
    t0 = time.perf_counter(); hip.hipMemset(p, 0xA5, C.c_size_t(n)); hip.hipDeviceSynchronize(); t_set = time.perf_counter() - t0
t_free = 0.0
t_again = 0.0
if free:
    t0 = time.perf_counter(); hip.hipFree(p); t_free = time.perf_counter() - t0
    q = C.c_void_p()
    t0 = time.perf_counter(); hip.hipMalloc(C.byref(q), C.c_size_t(n)); t_again = time.perf_counter() - t0
print("rc %d  context %.3f s  hipMalloc(%.0f GB) %.4f s  memset %.3f s  hipFree %.4f s  hipMalloc again (same process) %.4f s" % (rc, t_ctx, gb, t_malloc, t_set, t_free, t_again), flush=True)
'''


def child(gb, dirty, free, label):
    t = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", CHILD, str(gb), str(int(dirty)), str(int(free))], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print("%-58s %s  (process %.2f s)" % (label, r.stdout.strip() or r.stderr[-300:], time.perf_counter() - t), flush=True)


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 58
    child(gb, False, False, "1. first process, memory not written, exits without free:")
    child(gb, False, False, "2. next process at once:")
    child(gb, True, False, "3. next at once, WRITES all of it, exits without free:")
    child(gb, False, False, "4. next at once (after a writer):")
    child(gb, True, True, "5. next at once, writes, frees, allocates again:")
    time.sleep(5)
    child(gb, False, False, "6. five seconds later (after a writer):")
    child(gb, True, False, "7. next at once, writes:")
    time.sleep(5)
    child(gb, True, False, "8. five seconds later, writes:")
    child(gb / 8, True, False, "9. next at once, an eighth of the bytes, writes:")
    child(gb / 8, True, False, "10. next at once, an eighth, writes:")
    child(gb, False, False, "11. next at once, all of it:")
    for k in range(4):      # does the dirty state persist across several allocations?  (four writers in a row cover > 200 GB of the 288)
        child(gb, True, False, "12.%d writer in a row:" % k)


if __name__ == "__main__":
    main()
