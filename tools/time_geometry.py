"""count + lookups of the bench workload on tables of a given geometry (log2 slots, log2 bucket slots), for rocprofv3:
   python tools/time_geometry.py 29 14 29 13"""
import sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pangaea_amd import kmer, synth
dev = 'cuda:0'
cfg = synth.SynthConfig(n_pairs=10_000_000, n_barcodes=50_000, seed=2022)
s = synth.generate(cfg, device=dev, chunk_pairs=1 << 17, with_names=False)
rows = s.rows(2000); plan = kmer.Plan(rows, dev)
args = [int(a) for a in sys.argv[1:]] or [29, 14]
for ls, lb in zip(args[0::2], args[1::2]):
    t = kmer.KmerTable.mini_with_slots(21, dev, ls, lb)
    for it in range(3):
        t.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        t.count(s, rows=plan, emit=(10, 400), check=False)
        torch.cuda.synchronize()
        print(ls, lb, 'count+lookups (first call includes the plan) ms', round(1e3 * (time.perf_counter() - t0), 2), flush=True)
