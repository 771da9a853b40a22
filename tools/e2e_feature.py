#!/usr/bin/env python3
"""end-to-end timing of the feature step from a FASTQ file: ingest -> H2D -> table + rows -> D2H -> cache files"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pangaea_amd import kmer, synth  # noqa: E402
from pangaea_amd.feature import Feature, frame_like_read_csv, write_csv_gz  # noqa: E402
from pangaea_amd.reads import ReadStream  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tmp = tempfile.mkdtemp(prefix="pg_e2e_")
fq = os.path.join(tmp, "reads.fq")
cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=n_pairs // 200, seed=5)
t0 = time.perf_counter()
synth.write_fastq(synth.generate(cfg, device="cuda:0", chunk_pairs=1 << 17), cfg, fq)
print(f"(wrote {os.path.getsize(fq) / 1e6:.0f} MB FASTQ in {time.perf_counter() - t0:.1f} s)")
torch.cuda.synchronize()
kmer.count_kmers(synth.generate(synth.SynthConfig(n_pairs=2000, n_barcodes=10), device="cuda:0"), 21)   # warm the runtime


def lap(msg, t):
    torch.cuda.synchronize()
    now = time.perf_counter()
    print(f"  {msg:34s} {1e3 * (now - t):8.1f} ms")
    return now


torch.cuda.empty_cache()          # (the generator's temporaries: otherwise the first big allocation pays for flushing them)
t = t_all = time.perf_counter()
host = ReadStream.from_fastq(fq); t = lap("ingest (threaded parse + pack)", t)
s = host.to("cuda:0"); t = lap("H2D (pageable)", t)
rows = s.rows(2000); plan = kmer.Plan(rows, "cuda:0"); t = lap("rows + plan", t)
table = kmer.count_kmers(s, 21, rows=plan); t = lap("K2 table", t)
tnf, abd = kmer.features(s, plan, k_tnf=4, table=table); t = lap("K1 + K3 rows", t)
tnf_h, abd_h = tnf.cpu().numpy(), abd.cpu().numpy(); t = lap("D2H", t)
write_csv_gz(os.path.join(tmp, "t.gz"), rows.names, tnf_h); write_csv_gz(os.path.join(tmp, "a.gz"), rows.names, abd_h); t = lap("CSV.gz caches", t)
frame_like_read_csv(rows.names, tnf_h).to_pickle(os.path.join(tmp, "t.pkl")); frame_like_read_csv(rows.names, abd_h).to_pickle(os.path.join(tmp, "a.pkl")); t = lap("pickle caches", t)
total = time.perf_counter() - t_all
print(f"total {total:.2f} s -> {n_pairs / total / 1e6:.2f} M pairs/s end to end ({len(rows)} rows)")
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
