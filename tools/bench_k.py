import sys, time, torch
sys.path.insert(0, "/root/repo")
from pangaea_amd import kmer, synth
cfg = synth.SynthConfig(n_pairs=10_000_000, n_barcodes=50_000, seed=2022)
s = synth.generate(cfg, device="cuda:0", chunk_pairs=1 << 17, with_names=False)
rows = s.rows(2000); plan = kmer.Plan(rows, "cuda:0")
for k, kind in ((15, "dense"), (15, "hash"), (11, "dense"), (11, "hash")):
    t = kmer.KmerTable.alloc(k, "cuda:0", kind, distinct_hint=200_000_000 if k > 12 else 4 ** k)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        t.reset(); t.count(s, rows=plan, check=False); torch.cuda.synchronize(); t1 = time.perf_counter()
        tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=10, vsize=400); torch.cuda.synchronize(); t2 = time.perf_counter()
    t.check_status()
    print(f"k={k} {kind:5s} bucket={t.log2_bucket}: count {1e3*(t1-t0):6.1f} ms  features {1e3*(t2-t1):6.1f} ms  sum(abd)={int(abd.sum())}")
    del t
