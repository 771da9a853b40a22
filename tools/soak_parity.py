#!/usr/bin/env python3
"""Randomised parity soak on one GPU box: the super-k-mer pipeline (table + abundance rows from its emitted words) against the
key-partitioned kernels (hash / wide table + rows by lookups) on random geometries -- k, table and bucket sizes from one bucket to
2^16, window, vector size, row cut-off, 200 to 300 k pairs.  usage: python tools/soak_parity.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pangaea_amd import kmer, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = "cuda:0"
t_end = time.time() + budget
n = 0
while time.time() < t_end:
    k = int(rng.choice([13, 14, 15, 16, 17, 19, 20, 21, 21, 21, 22, 25, 27, 31]))
    n_pairs = int(np.exp(rng.uniform(np.log(200), np.log(300_000))))
    lb = int(rng.randint(8, 15)) if k <= 21 else int(rng.randint(8, 14))
    need = max(lb, int(np.ceil(np.log2(max(2, n_pairs * 2 * 130 / 0.5)))))            # room for every k-mer at load <= 0.5
    log2_slots = int(min(lb + 16, need + rng.randint(0, 2)))
    if log2_slots < need:
        lb = min(14 if k <= 21 else 13, need - 16 if need - 16 > lb else lb); log2_slots = max(need, lb)
    if rng.rand() < 0.3:                                     # tables of 2^16 buckets (the 512-digit second pass), whatever the data need
        lb = int(rng.randint(8, 13)); log2_slots = lb + 16
    window, vsize = int(rng.choice([1, 2, 3, 10, 25])), int(rng.choice([6, 50, 64, 400, 512]))
    min_len = int(rng.choice([0, 302, 600, 2000]))
    cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=max(1, n_pairs // int(rng.choice([7, 20, 200]))), n_genomes=3, genome_len=int(rng.choice([20_000, 300_000])),
                            fragment=8_000, sub_rate=0.01, n_rate=float(rng.choice([0.0, 0.2])), seed=int(rng.randint(1 << 30)))
    s = synth.generate(cfg, device=dev, with_names=False)
    rows = s.rows(min_len)
    if len(rows) == 0 or len(rows) >= (1 << 20):
        continue
    plan = kmer.Plan(rows, dev)
    tag = f"k={k} pairs={n_pairs} slots=2^{log2_slots} bucket=2^{lb} w={window} v={vsize} minlen={min_len} rows={len(rows)} seed={cfg.seed}"
    try:
        t = kmer.KmerTable.mini_with_slots(k, dev, log2_slots, lb)
        t.count(s, rows=plan, emit=(window, vsize))
        _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=window, vsize=vsize)
        h = kmer.count_kmers(s, k, kind="hash" if k <= 21 else "wide")
        _, want = kmer.features(s, rows, k_tnf=None, table=h, window=window, vsize=vsize)
        ok = torch.equal(abd, want)
        if ok and n_pairs <= 20_000:
            a, b = t.items(), h.items()
            ok = all(np.array_equal(x, y) for x, y in zip(a, b))
    except Exception as e:
        if "is full" in str(e):                              # (a fixed geometry with tiny buckets may overflow one: reported, not wrong)
            print("full ", tag, flush=True)
            continue
        print("ERROR", tag, repr(e)[:200], flush=True)
        raise
    n += 1
    print(("ok   " if ok else "FAIL ") + tag, flush=True)
    if not ok:
        sys.exit(1)
    del s, t, h, abd, want
print(f"soak: {n} cases, all equal")
