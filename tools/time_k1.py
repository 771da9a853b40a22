"""K1 (TNF rows) of the bench workload for several segment lengths of the row plan, for rocprofv3 or wall time"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pangaea_amd import kmer, synth
dev = 'cuda:0'
cfg = synth.SynthConfig(n_pairs=10_000_000, n_barcodes=50_000, seed=2022)
s = synth.generate(cfg, device=dev, chunk_pairs=1 << 17, with_names=False)
rows = s.rows(2000)
ref = None
for sc in [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536, 1 << 20]:
    plan = kmer.Plan(rows, dev, sc)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tnf, _ = kmer.features(s, plan, k_tnf=4)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ref = tnf if ref is None else ref
    print('seg_chars', sc, 'segments', plan.n_segs, 'ms', round(1e3 * dt, 3), 'same', bool(torch.equal(tnf, ref)), flush=True)
