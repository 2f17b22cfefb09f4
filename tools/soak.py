"""Developer soak (GPU box): random dataset shapes against the C oracle, every batch run three times (the self-resetting
device counters -- split tickets, dynamic work items -- must leave no state behind), on the path the batch chooses and on the
other one; the same work cut into pieces (midas_snps_contigs.origin); the rows from the device's coder against the host
formatter's text; the columns' bytes through zlib and back through the device inflater; the reads as a BAM decoded resident on the
device and batched in place (round 6).
usage: python tools/soak.py [seconds] [seed] [big]"""
import gzip
import os
import sys
import tempfile
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midas_amd import abi, pieces, synth  # noqa: E402
from oracle import c_oracle  # noqa: E402


def hot_spot(contigs, reads, rng, share):
    """Move `share` of the reads of the first contig into a 600-site window (keeps them sorted)."""
    n0 = int(contigs.read_begin[1])
    if n0 < 50:
        return
    pick = rng.random(n0) < share
    span = max(1, min(600, int(contigs.length[0]) - int(reads.l_seq.max()) - 40))
    pos = reads.pos[:n0].copy()
    pos[pick] = 10 + (rng.random(int(pick.sum())) * span).astype(np.int32)
    order = np.argsort(pos, kind='stable')
    sub = synth.take_reads(reads, np.concatenate([order, np.arange(n0, reads.n_reads)]))
    sub.pos[:n0] = pos[order]
    return sub


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
    ctx = abi.Context(0)
    t_end = time.time() + budget
    n = bad = 0
    tally = {'rows': 0, 'pieces': 0, 'inflate': 0, 'resident': 0, 'streamed': 0}
    while time.time() < t_end:
        read_len = int(rng.choice([36, 75, 100, 125, 150, 151, 250]))
        big = len(sys.argv) > 3 and sys.argv[3] == 'big'      # > 1024 tiles: the dynamic work-item path
        contig_len = int(rng.integers(400000, 900000)) if big else int(rng.integers(read_len + 40, 120000))
        ncontig = int(rng.integers(4, 12)) if big else int(rng.integers(1, 12))
        nsp = int(rng.integers(1, 4))
        cov = float(rng.choice([0.0, 0.3, 2, 8, 25, 60]))
        n_reads = int(min(2500000 if big else 400000, cov * contig_len * ncontig * nsp / read_len))
        kw = dict(n_species=nsp, contigs_per_species=ncontig, contig_len=contig_len, n_reads=max(n_reads, 0), read_len=read_len,
                  seed=int(rng.integers(1, 1 << 30)), var_len=bool(rng.random() < 0.5), lowercase_frac=0.05)
        contigs, reads = synth.make_dataset(**kw)
        tag = ''
        if reads.n_reads > 3000 and rng.random() < 0.4:
            moved = hot_spot(contigs, reads, rng, float(rng.choice([0.3, 0.9])))
            if moved is not None:
                reads, tag = moved, ' hot'
        if 200 < reads.n_reads <= 150000 and not tag and rng.random() < 0.35:
            # outliers: a few reads get a long deletion / skip in the middle (their span leaves the overhang, or several tiles)
            cig, coff = reads.cigar.copy(), reads.cigar_off
            pick = rng.choice(reads.n_reads, size=min(40, reads.n_reads // 50 + 1), replace=False)
            new_cig, new_off = [], [0]
            chosen = set(int(x) for x in pick)
            for i in range(reads.n_reads):
                ops = cig[coff[i]:coff[i + 1]]
                if i in chosen and ops.size == 1 and (ops[0] & 15) == 0 and (ops[0] >> 4) >= 20:
                    l = int(ops[0] >> 4)
                    a_ = int(rng.integers(5, l - 5))
                    gap = int(rng.choice([12, 40, 300, 2100, 5000]))
                    ops = np.array([(a_ << 4) | 0, (gap << 4) | int(rng.choice([2, 3])), ((l - a_) << 4) | 0], np.uint32)
                new_cig.append(ops)
                new_off.append(new_off[-1] + ops.size)
            reads = abi.ReadsSoA(**{**reads.as_dict(), "cigar": np.concatenate(new_cig) if new_cig else cig, "cigar_off": np.array(new_off, np.int64)})
            tag = ' outliers'
        args = dict(abi.DEFAULT_ARGS)
        if rng.random() < 0.3:
            args.update(baseq=int(rng.choice([0, 20, 41])), mapq=int(rng.choice([0, 30])), mapid=float(rng.choice([90.0, 97.5])),
                        aln_cov=float(rng.choice([0.5, 0.95])), readq=int(rng.choice([0, 30])))
        thr = abi.Thresholds.from_args(args)
        st, er, oc, oa, os_ = c_oracle.pileup(thr, contigs, reads)
        b = ctx.batch(contigs, reads)
        ok = True
        for rep in range(3):
            b.run(thr)
            counts, allele, stats = b.fetch()
            ok = ok and np.array_equal(counts, oc) and np.array_equal(allele, oa) and np.array_equal(stats, os_)
        info = b.info()
        extra = []
        if st == 0 and ok:
            # the other path
            other = abi.PATH_PACKED if info.path == abi.PATH_DIRECT else abi.PATH_DIRECT
            try:
                b.select_path(other)
                b.run(thr)
                c2, a2, s2 = b.fetch()
                if not (np.array_equal(c2, oc) and np.array_equal(a2, oa) and np.array_equal(s2, os_)):
                    ok = False; extra.append("other path")
            except abi.MidasSnpsError:
                pass                                   # (the direct path declines unsorted input; there is none here)
            # the long path (pileup_long.hip): a third implementation of the same rules (small cases: it is built for exactness)
            if info.path != abi.PATH_LONG and reads.n_reads <= 400000:
                b.select_path(abi.PATH_LONG)
                b.run(thr)
                c4, a4, s4 = b.fetch()
                if not (np.array_equal(c4, oc) and np.array_equal(a4, oa) and np.array_equal(s4, os_)):
                    ok = False; extra.append("long path")
                b.select_path(abi.PATH_AUTO)
            # rows: device coder vs host formatter, as text
            if contigs.n_sites <= 3000000 and rng.random() < 0.5:
                with tempfile.TemporaryDirectory() as td:
                    pick = list(range(contigs.n_contigs))
                    ctx.set_row_coder(abi.ROWS_DEVICE)
                    b.write_part(td + "/d.gz", pick, contigs.ids, header=True, gz_level=4, threads=4)
                    off = contigs.site_offsets()
                    abi.write_table(td + "/h.gz", contigs.ids, [oa[off[k]:off[k + 1]] for k in pick],
                                    [oc[off[k]:off[k + 1]] for k in pick], gz_level=4, threads=4)
                    if gzip.open(td + "/d.gz", "rb").read() != gzip.open(td + "/h.gz", "rb").read():
                        ok = False; extra.append("rows")
                    tally['rows'] += 1
        b.close()
        if st == 0 and ok and reads.n_reads > 0 and not tag and rng.random() < 0.5:
            # the same work in pieces
            pt, pr, _ = pieces.split_table(contigs, reads, int(rng.choice([65536, 131072, 262144])))
            if pt.n_contigs > contigs.n_contigs:
                c3, a3, s3 = ctx.pileup(thr, pt, pr)
                if not (np.array_equal(c3, oc) and np.array_equal(a3, oa) and np.array_equal(s3, os_)):
                    ok = False; extra.append("pieces")
                tally['pieces'] += 1
        if reads.n_reads > 0 and rng.random() < 0.3:
            # BAM-like bytes through zlib and back through the device inflater, in BGZF-sized streams
            blob = np.concatenate([reads.seq4[:400000], reads.qual[:400000], reads.cigar[:50000].view(np.uint8)]).tobytes()
            chunks = [blob[i:i + 65280] for i in range(0, len(blob), 65280)]
            lvl = int(rng.choice([1, 6, 9]))
            streams = []
            for ch in chunks:
                co = zlib.compressobj(lvl, zlib.DEFLATED, -15)
                streams.append(co.compress(ch) + co.flush())
            cpos = np.concatenate([[0], np.cumsum([len(s) for s in streams])])[:-1]
            upos = np.concatenate([[0], np.cumsum([len(c) for c in chunks])])[:-1]
            got = ctx.inflate_blocks(b"".join(streams), cpos, [len(s) for s in streams], upos, [len(c) for c in chunks], len(blob))
            if bytes(got) != blob:
                ok = False; extra.append("inflate")
            tally['inflate'] += 1
        if st == 0 and ok and reads.n_reads > 0 and reads.n_reads <= 600000 and rng.random() < 0.5:
            # ONE pass from the BAM's bytes: the same reads written as a BAM, decoded on the device into the kernel's own layout,
            # batched where they lie (midas_snps_batch_create_resident) -- the direct path, then the packed or the long one
            # (their columns cut out of the handle's stream)
            with tempfile.TemporaryDirectory() as td:
                path = td + "/s.bam"
                refid = np.repeat(np.arange(contigs.n_contigs, dtype=np.int32), np.diff(contigs.read_begin))
                abi.write_bam(path, contigs.ids, [int(x) for x in contigs.length], refid, reads, level=int(rng.choice([1, 6])))
                # ... half of them through the STREAMED decode (groups of a few blocks, one to three slots: records straddle every
                # group's end, the raw columns of the packed / long path are cut out of the direct layout)
                streamed = rng.random() < 0.5
                if streamed:
                    os.environ["MIDAS_SNPS_DECODE_GROUP_BLOCKS"] = str(max(int(rng.integers(1, 40)), os.path.getsize(path) // (20000 * 24)))     # (<= ~30 groups: a group's decoder takes 25 ms however small)
                    os.environ["MIDAS_SNPS_DECODE_SLOTS"] = str(int(rng.integers(1, 4)))
                    tally['streamed'] += 1
                else:
                    os.environ["MIDAS_SNPS_DECODE_STREAM"] = "0"
                try:
                    _, _, rid, res = abi.read_bam(path, ctx, resident=True)
                except abi.MidasSnpsError as e:
                    print("DECODE FAILED", {k: os.environ.get(k) for k in ("MIDAS_SNPS_DECODE_GROUP_BLOCKS", "MIDAS_SNPS_DECODE_SLOTS", "MIDAS_SNPS_DECODE_STREAM")},
                          os.path.getsize(path), e.message, flush=True)
                    raise
                finally:
                    for k in ("MIDAS_SNPS_DECODE_GROUP_BLOCKS", "MIDAS_SNPS_DECODE_SLOTS", "MIDAS_SNPS_DECODE_STREAM"):
                        os.environ.pop(k, None)
                rb = ctx.batch(contigs, res)
                for pth in (None, abi.PATH_PACKED if rng.random() < 0.5 else abi.PATH_LONG):
                    try:
                        if pth is not None:
                            rb.select_path(pth)
                        rb.run(thr)
                        c5, a5, s5 = rb.fetch()
                        if not (np.array_equal(c5, oc) and np.array_equal(a5, oa) and np.array_equal(s5, os_)):
                            ok = False; extra.append("resident %s" % abi.PATH_NAMES[rb.info().path])
                    except abi.MidasSnpsError as e:
                        if pth is None:
                            ok = False; extra.append("resident: %s" % e.message)
                rb.close()
                del res, rid
            tally['resident'] += 1
        n += 1
        if not ok:
            bad += 1
            print("MISMATCH", kw, tag, args, extra, flush=True)
        elif n % 10 == 0:
            print("%d cases ok (last: %d sites, %d reads, %d tiles, %d items%s)" % (n, info.n_sites, info.n_reads, info.n_tiles,
                                                                                   info.n_work_items, tag), flush=True)
    print("soak: %d cases (of them %s), %d mismatches" % (n, ", ".join("%d with %s" % (v, k) for k, v in tally.items()), bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
