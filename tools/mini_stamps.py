import sys, torch
sys.path.insert(0, '.')
from pangaea_amd import kmer, synth
dev = 'cuda:0'
cfg = synth.SynthConfig(n_pairs=10_000_000, n_barcodes=50_000, seed=2022)
s = synth.generate(cfg, device=dev, chunk_pairs=1 << 17, with_names=False)
rows = s.rows(2000); plan = kmer.Plan(rows, dev)
t = kmer.KmerTable.mini_with_slots(21, dev, 29, 14)
import os
for it in range(2):
    t.reset(); t.count(s, rows=plan, emit=(10, 400), check=False)
    torch.cuda.synchronize()
    h = t._mini_plan[1][:512].view(torch.int64).cpu().numpy()
    print('records', h[0], 'words', h[1], 'phase cycle sums (per WG avg, cycles at 100MHz?):', [int(x) // 32768 for x in h[8:12]], 'wave-end avg', int(h[13]) // (32768 * 16), 'of which general insert', int(h[14]) // (32768 * 16), 'lookup phase per WG (zero+barrier, load+bin+rank, scan, cursor+place, copy-out):', [int(x) // 32768 for x in h[24:29]])
    print('  count loop per wave (top wait, codes+probes, hits+claim, ring+general, word stores):', [int(x) // (32768 * 16) for x in h[40:45]])
    if h[62]:
        print('  merged lookup per WG (stage A: loads, bins, runs, appends | stage B: ranks, cursor adds, scan, placement | copy-out):', [int(x) // 32768 for x in h[56:59]], 'sort rounds per WG', round(h[62] / 32768, 2), 'words per round', int(h[63] // max(1, h[62])), 'merged words / provisional words', round(h[63] / max(1, h[9 + 0] * 0 + h[1]), 3) if h[1] else None)
    if h[18]:
        print('  general form: words settled by the first probe', h[18], 'runs of equal neighbours inside their records', h[19], 'ratio', round(h[19] / h[18], 3))
    t._mini_plan[1][64:512].zero_()
