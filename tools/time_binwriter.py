#!/usr/bin/env python3
"""time the bin writer (pg_extract_reads: clusters.tsv -> cluster_bin<label>.{fq,barcode}) on a synthetic interleaved FASTQ"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pangaea_amd import synth  # noqa: E402
from pangaea_amd.binwriter import extract_reads  # noqa: E402
from pangaea_amd.clustering import write_clusters_tsv  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
clusters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tmp = tempfile.mkdtemp(prefix="pg_bins_")
fq = os.path.join(tmp, "reads.fq")
cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=max(1, n_pairs // 200), seed=5)
stream = synth.generate(cfg)
synth.write_fastq(stream, cfg, fq)
names = sorted({n for n in stream.run_names if n})
labels = np.random.RandomState(0).randint(-1, clusters, len(names))
tsv = os.path.join(tmp, "clusters.tsv")
write_clusters_tsv(tsv, labels, names)
t = time.perf_counter()
written = extract_reads(fq, None, tsv, os.path.join(tmp, "cluster"))
dt = time.perf_counter() - t
size = os.path.getsize(fq)
print(f"{n_pairs} pairs, {size / 1e6:.0f} MB, {clusters} clusters: {written} pairs written in {dt:.2f} s = {n_pairs / dt / 1e6:.2f} M pairs/s, {size / dt / 1e9:.2f} GB/s read")
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
