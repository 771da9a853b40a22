#!/usr/bin/env python3
"""time pg_ingest_fastq on a synthetic interleaved FASTQ, and on the same reads as an R1 / R2 pair of files, for several
thread counts (host-only; no GPU work)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pangaea_amd import _lib, synth  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
path = "/tmp/pg_ingest_test.fq"
cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=max(1, n_pairs // 200))
synth.write_fastq(synth.generate(cfg), cfg, path)
L = _lib.load()
print(f"{n_pairs} pairs, {os.path.getsize(path) / 1e6:.0f} MB, nproc {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}")
for T in (1, 2, 4, 8, 16, 32):
    L.pg_set_ingest_threads(T)
    best = 1e9
    for _ in range(3):
        h = C.c_void_p()
        t = time.perf_counter()
        _lib.check(L.pg_ingest_fastq(path.encode(), None, C.byref(h)))
        best = min(best, time.perf_counter() - t)
        L.pg_reads_free(h)
    print(f"threads {T:2d}: {best * 1e3:7.1f} ms  {n_pairs / best / 1e6:6.2f} M pairs/s")
# the same reads as -1 / -2 files
p1, p2 = path + ".1", path + ".2"
with open(path) as f, open(p1, "w") as o1, open(p2, "w") as o2:
    while True:
        rec = [f.readline() for _ in range(8)]
        if not rec[0]:
            break
        o1.writelines(rec[:4]); o2.writelines(rec[4:])
print("paired files (-1 / -2):")
for T in (1, 4, 16, 32):
    L.pg_set_ingest_threads(T)
    best = 1e9
    for _ in range(2 if T == 1 else 3):
        h = C.c_void_p()
        t = time.perf_counter()
        _lib.check(L.pg_ingest_fastq(p1.encode(), p2.encode(), C.byref(h)))
        best = min(best, time.perf_counter() - t)
        L.pg_reads_free(h)
    print(f"threads {T:2d}: {best * 1e3:7.1f} ms  {n_pairs / best / 1e6:6.2f} M pairs/s")
# gzip (level 1, as sequencers and pigz -1 write it): one inflate stream per file, then the threaded parse
import gzip  # noqa: E402
import shutil  # noqa: E402
for q in (path, p1, p2):
    with open(q, "rb") as fi, gzip.open(q + ".gz", "wb", compresslevel=1) as fo:
        shutil.copyfileobj(fi, fo, 1 << 24)
for label, a, b in (("interleaved .gz", path + ".gz", None), ("paired .gz", p1 + ".gz", p2 + ".gz")):
    for T in (1, 32):
        L.pg_set_ingest_threads(T)
        h = C.c_void_p()
        t = time.perf_counter()
        _lib.check(L.pg_ingest_fastq(a.encode(), b.encode() if b else None, C.byref(h)))
        dt = time.perf_counter() - t
        L.pg_reads_free(h)
        print(f"{label:16s} threads {T:2d}: {dt * 1e3:7.1f} ms  {n_pairs / dt / 1e6:6.2f} M pairs/s")
for q in (path, p1, p2):
    os.remove(q)
    os.remove(q + ".gz")
