#!/usr/bin/env python3
"""Exercise the HOST half of the C ABI (no kernels) under a sanitizer build: run by tools/sanitize_host.sh"""
import ctypes as C
import glob
import gzip
import os
import sys
import tempfile

import numpy as np

lib = C.CDLL(sys.argv[1])
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
i64, vp, cp = C.c_int64, C.c_void_p, C.c_char_p
for f in ("pg_reads_n_words", "pg_reads_n_runs", "pg_reads_rows", "pg_reads_n_chars", "pg_plan_segments", "pg_words_for"):
    getattr(lib, f).restype = i64
for f in ("pg_reads_codes", "pg_reads_valid", "pg_reads_run_off"):
    getattr(lib, f).restype = vp
lib.pg_reads_run_name.restype = cp
lib.pg_last_error.restype = cp

tmp = tempfile.mkdtemp()
# a larger interleaved file so the threaded parser takes its path (>= 64 KiB per thread)
big = os.path.join(tmp, "big.fq")
with open(big, "w") as out:
    rng = np.random.RandomState(1)
    for b in range(400):
        bc = "".join("ACGT"[x] for x in rng.randint(0, 4, 12))
        for p in range(rng.randint(1, 9)):
            for m in (1, 2):
                seq = "".join("ACGTN"[x] for x in rng.randint(0, 5, rng.randint(30, 160)))
                out.write(f"@r{b}_{p} BX:Z:{bc}-1\n{seq}\n+\n{'I' * len(seq)}\n")
# the same reads as two files (-1 / -2) for the threaded paired reader: some names that do not match, qualities below '?', a
# quality line one short, lower-case bases, R2 a few records longer, no newline at the end of R1
big1, big2 = os.path.join(tmp, "big_1.fq"), os.path.join(tmp, "big_2.fq")
recs = open(big).read().splitlines()
recs = [recs[i:i + 4] for i in range(0, len(recs), 4)]
with open(big1, "w") as o1, open(big2, "w") as o2:
    for i, (a, b) in enumerate(zip(recs[0::2], recs[1::2])):
        a[3] = "".join("#5?I"[x] for x in rng.randint(0, 4, len(a[1])))
        b[3] = b[3][:len(b[3]) - (i % 5 == 0)]
        if i % 9 == 0:
            b[0] = b[0].replace("@r", "@q", 1)
        if i % 4 == 0:
            a[1] = a[1][:7].lower() + a[1][7:]
        o2.write("\n".join(b) + "\n")
        if i < len(recs) // 2 - 6:
            o1.write("\n".join(a) + ("\n" if i < len(recs) // 2 - 7 else ""))
inputs = [(os.path.join(G, f), None) for f in ("tenx_mixed.fq", "tenx_clean.fq.gz", "stlfr.fq", "tenx_crlf.fq", "tenx_single.fq", "polya.fq.gz")]
inputs += [(os.path.join(G, "pair_R1.fq"), os.path.join(G, "pair_R2.fq")), (big, None), (big1, big2)]
ref = {}
for threads in (1, 2, 5, 16):
    lib.pg_set_ingest_threads(threads)
    for r1, r2 in inputs:
        h = vp()
        rc = lib.pg_ingest_fastq(r1.encode(), r2.encode() if r2 else None, C.byref(h))
        assert rc == 0, lib.pg_last_error()
        nw, nr = lib.pg_reads_n_words(h), lib.pg_reads_n_runs(h)
        codes = np.ctypeslib.as_array(C.cast(lib.pg_reads_codes(h), C.POINTER(C.c_uint64)), (nw,)).copy()
        off = np.ctypeslib.as_array(C.cast(lib.pg_reads_run_off(h), C.POINTER(C.c_int64)), (nr + 1,)).copy()
        names = [lib.pg_reads_run_name(h, i) for i in range(nr)]
        rows = np.zeros(nr, np.int64)
        n_rows = lib.pg_reads_rows(h, 100, vp(rows.ctypes.data))
        key = (r1, r2)
        if key in ref:
            assert np.array_equal(ref[key][0], codes) and np.array_equal(ref[key][1], off) and ref[key][2] == names
        else:
            ref[key] = (codes, off, names)
        lib.pg_reads_free(h)
# sharded ingest of the plain files, also with tiny reader blocks (refills, buffer growth)
for block in (None, "16", "300"):
    if block:
        os.environ["PG_INGEST_BLOCK"] = block
    for threads in (1, 4):
        lib.pg_set_ingest_threads(threads)
        for path in (big, os.path.join(G, "tenx_mixed.fq"), os.path.join(G, "stlfr.fq")):
            for parts in (1, 3, 8):
                counts = []
                for r in range(parts):
                    c = i64(0)
                    assert lib.pg_fastq_count_newlines(path.encode(), r, parts, C.byref(c)) == 0, lib.pg_last_error()
                    counts.append(c.value)
                before = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
                names, total = [], 0
                for r in range(parts):
                    h = vp()
                    assert lib.pg_ingest_fastq_shard(path.encode(), r, parts, vp(before.ctypes.data), C.byref(h)) == 0, lib.pg_last_error()
                    names += [lib.pg_reads_run_name(h, i) for i in range(lib.pg_reads_n_runs(h))]
                    total += lib.pg_reads_n_chars(h)
                    lib.pg_reads_free(h)
                assert names == ref[(path, None)][2] and total == ref[(path, None)][1][-1]
os.environ.pop("PG_INGEST_BLOCK", None)
# error paths
h = vp()
assert lib.pg_ingest_fastq(b"/nonexistent", None, C.byref(h)) == -2
bad = os.path.join(tmp, "bad.fq")
open(bad, "w").write("@r BX:Z\nACGT\n+\nIIII\n" * 2)
assert lib.pg_ingest_fastq(bad.encode(), None, C.byref(h)) == -3
# pack / plan / colmap / csv / bin writer
text = bytes(np.random.RandomState(2).choice(list(b"ACGTNacgt"), 5001).astype(np.uint8))
nw = lib.pg_words_for(i64(len(text)))
codes, valid = np.zeros(nw, np.uint64), np.zeros(nw, np.uint32)
assert lib.pg_pack_ascii(text, i64(len(text)), vp(codes.ctypes.data), vp(valid.ctypes.data)) == 0
start = np.array([0, 100, 5000], np.int64); end = np.array([90, 4000, 5001], np.int64)
n = lib.pg_plan_segments(vp(start.ctypes.data), vp(end.ctypes.data), i64(3), i64(64), None, None, None)
sr, ss, se = np.zeros(n, np.int32), np.zeros(n, np.int64), np.zeros(n, np.int64)
lib.pg_plan_segments(vp(start.ctypes.data), vp(end.ctypes.data), i64(3), i64(64), vp(sr.ctypes.data), vp(ss.ctypes.data), vp(se.ctypes.data))
for k in range(1, 7):
    cm = np.zeros(4 ** k, np.uint16); cc = np.zeros(lib.pg_tnf_ncols(k), np.uint32)
    assert lib.pg_tnf_colmap(k, vp(cm.ctypes.data), vp(cc.ctypes.data)) > 0
mat = np.arange(12, dtype=np.int32).reshape(3, 4) * 400000
assert lib.pg_write_csv_gz(os.path.join(tmp, "m.gz").encode(), b"a\0b\0c\0", vp(mat.ctypes.data), i64(3), i64(4)) == 0
assert gzip.open(os.path.join(tmp, "m.gz")).read().startswith(b"a,0,400000,800000,1.2e+06")
tsv = os.path.join(tmp, "c.tsv")
open(tsv, "w").write("3\tAAACCCGG,ACGTACGT\n-1\tAACGTTTC\n0\tCCGGTTAA\n")
wrote = i64(0)
assert lib.pg_extract_reads(os.path.join(G, "tenx_mixed.fq").encode(), None, tsv.encode(), os.path.join(tmp, "cl").encode(), C.byref(wrote)) == 0
assert lib.pg_extract_reads(os.path.join(G, "pair_R1.fq").encode(), os.path.join(G, "pair_R2.fq").encode(), tsv.encode(), os.path.join(tmp, "cp").encode(), C.byref(wrote)) == 0
# the threaded bin writer on the larger file (sizes pass + pwrite pass), also with tiny reader blocks
tsv2 = os.path.join(tmp, "c2.tsv")
bcs = sorted({n.decode() for n in ref[(big, None)][2] if n})
open(tsv2, "w").write("".join(f"{i % 5}\t{','.join(bcs[i::7])}\n" for i in range(5)))
sizes = None
for threads, block in ((1, None), (4, None), (6, "300")):
    lib.pg_set_ingest_threads(threads)
    if block:
        os.environ["PG_INGEST_BLOCK"] = block
    assert lib.pg_extract_reads(big.encode(), None, tsv2.encode(), os.path.join(tmp, f"w{threads}").encode(), C.byref(wrote)) == 0, lib.pg_last_error()
    got = sorted((os.path.basename(f)[2:], open(f, "rb").read()) for f in glob.glob(os.path.join(tmp, f"w{threads}_bin*")))
    assert wrote.value > 0 and (sizes is None or got == sizes)
    sizes = got
os.environ.pop("PG_INGEST_BLOCK", None)
# ... and on the same reads as -1 / -2 files
sizes = None
for threads, block in ((1, None), (3, None), (5, "300")):
    lib.pg_set_ingest_threads(threads)
    if block:
        os.environ["PG_INGEST_BLOCK"] = block
    assert lib.pg_extract_reads(big1.encode(), big2.encode(), tsv2.encode(), os.path.join(tmp, f"p{threads}").encode(), C.byref(wrote)) == 0, lib.pg_last_error()
    got = sorted((os.path.basename(f)[2:], open(f, "rb").read()) for f in glob.glob(os.path.join(tmp, f"p{threads}_bin*")))
    assert wrote.value > 0 and (sizes is None or got == sizes)
    sizes = got
os.environ.pop("PG_INGEST_BLOCK", None)
for f in glob.glob(os.path.join(tmp, "*")):
    os.remove(f)
os.rmdir(tmp)
print("sanitize_host: all host entry points exercised")
