"""Where the count kernel's CU time goes BETWEEN its workgroups: a variant build (make -C pangaea_amd/csrc variant NAME=gap KFLAGS=-DPG_MINI_GAPS)
whose workgroups leave (start, end, hardware id, records) in the dead record buffer; per CU: busy time, the gaps between one
workgroup's end and the next one's start.  PANGAEA_LIB=build/libpangaea_feat_gap.so PANGAEA_ALLOW_VARIANT=1 python tools/wg_gaps.py"""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from pangaea_amd import kmer, synth
dev = 'cuda:0'
cfg = synth.SynthConfig(n_pairs=10_000_000, n_barcodes=50_000, seed=2022)
s = synth.generate(cfg, device=dev, chunk_pairs=1 << 17, with_names=False)
rows = s.rows(2000); plan = kmer.Plan(rows, dev)
t = kmer.KmerTable.mini_with_slots(21, dev, 29, 14)
for it in range(3):
    t.reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t.count(s, rows=plan, emit=(10, 400), check=False); e1.record()
    torch.cuda.synchronize()
nb = 1 << 15
g = t._mini_rec_ws[:nb * 32].view(torch.int64).cpu().numpy().reshape(nb, 4)
start, end, hw, nrec = g[:, 0], g[:, 1], g[:, 2], g[:, 3]
cu = ((hw >> 32) & 15) * 4096 + ((hw >> 8) & 0xff)          # xcc, (se, sh, cu) bits of HW_ID
xcc = (hw >> 32) & 15
dur = end - start
print('count() took', round(e0.elapsed_time(e1), 2), 'ms; CUs seen', len(np.unique(cu)), '; workgroups per XCD', np.bincount(xcc).tolist())
print('workgroup duration ticks: mean', dur.mean(), 'min', dur.min(), 'max', dur.max(), '; records per bucket mean', nrec.mean(), 'max', nrec.max())
busy, gaps, tails, spans, heads = [], [], [], [], []
for x in np.unique(xcc):                                     # (every XCD has its own clock)
    mx = xcc == x
    x0, x1 = start[mx].min(), end[mx].max()
    spans.append(x1 - x0)
    for c in np.unique(cu[mx]):
        m = cu == c
        o = np.argsort(start[m]); st, en = start[m][o], end[m][o]
        busy.append((en - st).sum() / (x1 - x0))
        gaps.extend((st[1:] - en[:-1]).tolist())
        heads.append((st[0] - x0) / (x1 - x0)); tails.append((x1 - en[-1]) / (x1 - x0))
gaps = np.array(gaps)
print('XCD spans, ticks:', spans)
print('per CU, fractions of its XCD span: busy mean', round(np.mean(busy), 4), 'min', round(np.min(busy), 4), '; idle before the first workgroup', round(np.mean(heads), 4), '; idle behind the last', round(np.mean(tails), 4), 'max', round(np.max(tails), 4))
print('gap between workgroups on a CU, ticks: mean', round(gaps.mean(), 1), 'median', np.median(gaps), 'p75', np.percentile(gaps, 75), 'p90', np.percentile(gaps, 90), 'max', gaps.max(), '= of a workgroup', round(gaps.mean() / dur.mean(), 4))
print('correlation of duration and records:', round(np.corrcoef(dur, nrec)[0, 1], 3), '; ticks per record', round(dur.sum() / nrec.sum(), 3))
