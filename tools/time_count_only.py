"""the count kernel alone on the bench workload: count + lookups (fused), count only (no rows), for rocprofv3"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pangaea_amd import kmer, synth
dev = 'cuda:0'
cfg = synth.SynthConfig(n_pairs=10_000_000, n_barcodes=50_000, seed=2022)
s = synth.generate(cfg, device=dev, chunk_pairs=1 << 17, with_names=False)
rows = s.rows(2000); plan = kmer.Plan(rows, dev)
t = kmer.KmerTable.mini_with_slots(21, dev, 29, 14)
for what in (sys.argv[1:] or ("fused", "count-only", "count-only-no-rows")):
    for it in range(3):
        t.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        if what == "fused":
            t.count(s, rows=plan, emit=(10, 400), check=False)
        elif what == "count-only":
            t.count(s, rows=plan, check=False)
        else:
            t.count(s, check=False)
        torch.cuda.synchronize()
        print(what, 'ms', round(1e3 * (time.perf_counter() - t0), 2), flush=True)
