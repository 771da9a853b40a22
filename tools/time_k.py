#!/usr/bin/env python3
"""count + abundance rows for several k on the bench workload (one GPU): the super-k-mer pipeline (mini / miniw tables) against
the direct kernels that 22 <= k <= 31 ran before.   python tools/time_k.py [pairs] [k ...]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from pangaea_amd import kmer, synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ks = [int(a) for a in sys.argv[2:]] or [21, 27, 31]
dev = "cuda:0"
s = synth.generate(synth.SynthConfig(n_pairs=pairs, n_barcodes=pairs // 200, seed=2022), device=dev, chunk_pairs=1 << 17, with_names=False)
rows = s.rows(2000)
plan = kmer.Plan(rows, dev)


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for k in ks:
    t = kmer.count_kmers(s, k, rows=plan, emit=(10, 400))
    abd = torch.zeros((len(rows), 400), dtype=torch.int32, device=dev)

    def step():
        t.reset().count(s, rows=plan, emit=(10, 400), check=False)
        kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400, out_abd=abd.zero_())
    ms = timed(step)
    want = abd.clone()
    line = f"k={k}: {t.kind} 2^{t.log2_slots} slots / buckets of 2^{t.log2_bucket}: count + rows {ms:.1f} ms ({pairs / ms / 1e3:.0f} M pairs/s)"
    del t
    if k > 21 and pairs <= 10_000_000:
        w = kmer.count_kmers(s, k, kind="wide")

        def step_w():
            w.reset().count(s, check=False)
            kmer.features(s, rows, k_tnf=None, table=w, window=10, vsize=400, out_abd=abd.zero_())
        ms_w = timed(step_w, 1)
        line += f"; direct wide table: {ms_w:.1f} ms; rows identical: {bool(torch.equal(abd, want))}"
        del w
    print(line, flush=True)
