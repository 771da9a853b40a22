#!/usr/bin/env python3
"""FASTQ file -> packed stream in HBM, two ways (one GPU box): host ingest + one copy (`pg_ingest_fastq`, `.to(device)`)
against the ingest with the copy inside (`pg_ingest_fastq_device`: pieces copied under the parse, placement by a kernel),
for several piece sizes.  PG_INGEST_TIMING=1 prints the phases."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pangaea_amd import synth  # noqa: E402
from pangaea_amd.reads import ReadStream  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=max(1, n_pairs // 200), read_len=150, seed=2022)
stream = synth.generate(cfg, dev, with_names=False)
tmp = tempfile.mkdtemp(prefix="pg_ing_")
fq = os.path.join(tmp, "reads.fq")
synth.write_fastq_fast(stream, cfg, fq, n_pairs)
del stream
print(f"{n_pairs} pairs, {os.path.getsize(fq) / 1e9:.2f} GB, {len(os.sched_getaffinity(0))} host threads available", flush=True)


def timed(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        s = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
        ref = (s.n_chars, s.n_pairs, int(s.codes[:1000].sum()))
        del s
    print(f"{label:42s} {best * 1e3:8.1f} ms  {n_pairs / best / 1e6:6.1f} M pairs/s   {ref}", flush=True)


os.environ["PANGAEA_INGEST_ON_HOST"] = "1"
timed("host ingest + copy", lambda: ReadStream.from_fastq(fq, device=dev))
os.environ.pop("PANGAEA_INGEST_ON_HOST")
for piece in (None, 4 << 20, 8 << 20, 32 << 20, 64 << 20):
    if piece:
        os.environ["PG_INGEST_PIECE"] = str(piece)
    timed(f"device ingest, pieces of {(piece or 16 << 20) >> 20} MiB", lambda: ReadStream.from_fastq(fq, device=dev))
os.environ.pop("PG_INGEST_PIECE", None)
# the same reads as -1 / -2 files (pg_ingest_fastq_pair_device: pieces of R1, three planes)
r1, r2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
with open(fq, "rb") as f, open(r1, "wb") as a, open(r2, "wb") as b:
    while True:
        chunk = [f.readline() for _ in range(8)]
        if not chunk[7]:
            break
        a.writelines(chunk[:4]); b.writelines(chunk[4:])
os.remove(fq)
os.environ["PANGAEA_INGEST_ON_HOST"] = "1"
timed("-1/-2: host ingest + copy", lambda: ReadStream.from_fastq(r1, r2, device=dev))
os.environ.pop("PANGAEA_INGEST_ON_HOST")
for piece in (None, 4 << 20, 32 << 20):
    if piece:
        os.environ["PG_INGEST_PIECE"] = str(piece)
    timed(f"-1/-2: device ingest, pieces of {(piece or 16 << 20) >> 20} MiB", lambda: ReadStream.from_fastq(r1, r2, device=dev))
os.remove(r1); os.remove(r2)
os.rmdir(tmp)
