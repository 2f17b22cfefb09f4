"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/profile.sh into one text + JSON summary.
usage: python tools/summarize_prof.py gpurun_out/prof_<tag> > profiles/<name>.txt
"""
import glob
import re
import json
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    out = {"kernels": {}, "pmc": {}}
    tr = os.path.join(root, "trace", "trace_results.db")
    if os.path.exists(tr):
        cur = sqlite3.connect(tr).cursor()
        print("== rocprofv3 --kernel-trace --stats (%s) ==" % tr)
        print("%-90s %6s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
            print("%-90s %6d %12.3f %10.3f %7.2f" % (name[:90], calls, total, avg, pct))
            out["kernels"][name] = {"calls": calls, "total_us": total, "avg_us": avg, "pct": pct}
        q = "select name, min(duration), max(duration), vgpr_count, sgpr_count, lds_size, grid_x, workgroup_x from kernels group by name"
        for r in cur.execute(q):
            print("   %s  min %.1f us max %.1f us vgpr %s sgpr %s lds %s grid %s wg %s" % (
                r[0][:60], r[1] / 1e3, r[2] / 1e3, r[3], r[4], r[5], r[6], r[7]))
    print()
    print("== rocprofv3 --pmc passes (per-dispatch averages) ==")
    for db in sorted(glob.glob(os.path.join(root, "pmc_*", "pmc_results.db"))):
        cur = sqlite3.connect(db).cursor()
        q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
             "where kernel_name like '%midas%' or kernel_name like '%merge_sites%' group by kernel_name, counter_name")
        for k, c, v, n in cur.execute(q):
            m = re.search(r"(\w+_kernel(?:<\w+>)?)", k)
            short = m.group(1) if m else k[:40]
            print("%-22s %-28s %18.1f  (n=%d)" % (short, c, v, n))
            out["pmc"].setdefault(short, {})[c] = v
    step = [k for k in out["pmc"] if k.startswith(("direct_ranges_kernel", "pileup_direct_kernel"))]
    for k, d in out["pmc"].items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            # MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
            # half the bytes of a wide coalesced streaming read -> double it.
            rd = d["FETCH_SIZE"] * 1024 * 2
            wr = d["WRITE_SIZE"] * 1024
            d["hbm_read_bytes_corrected"] = rd
            d["hbm_write_bytes"] = wr
            d["hbm_bytes_per_launch"] = rd + wr
            print("%-22s HBM traffic/launch: read %.1f MB (FETCH_SIZE KiB x1024 x2 gfx950 correction) + write %.1f MB = %.1f MB"
                  % (k, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
    if step and all("hbm_bytes_per_launch" in out["pmc"][k] for k in step):
        tot = sum(out["pmc"][k]["hbm_bytes_per_launch"] for k in step)
        out["direct_step_hbm_bytes"] = tot
        print("%-22s HBM traffic/step (direct_ranges_kernel + pileup_direct_kernel; FETCH_SIZE x2: the factor bench.py calibrates on the box for 4-, 8- and 16-byte loads): %.1f MB" % ("direct step", tot / 1e6))
    print()
    print("JSON:", json.dumps(out))


if __name__ == "__main__":
    main()
