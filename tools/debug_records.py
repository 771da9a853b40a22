import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pangaea_amd import kmer, synth, _lib
DEV = "cuda:0"
cfg = synth.SynthConfig(n_pairs=300, n_barcodes=10, n_genomes=3, genome_len=30_000, fragment=8_000, sub_rate=0.01, n_rate=0.2, seed=521)
s = synth.generate(cfg, device=DEV)
rows = s.rows(2000); plan = kmer.Plan(rows, DEV)
t = kmer.KmerTable.mini_with_slots(21, DEV, 19, 14)
t.count(s, rows=plan, emit=(10, 400), check=False)
torch.cuda.synchronize()
print("status", t.status.cpu().numpy())
nrec, nlong = t.plan_counts()
ws = t._mini_rec_ws
cap = ws.numel() // 24 // 256 * 256
bases = ws[:8 * cap].view(torch.int64)[:nrec].cpu().numpy().view(np.uint64)
meta = ws[16 * cap:20 * cap].view(torch.int32)[:nrec].cpu().numpy().view(np.uint32)
n = ((meta >> 7) & 15) + 1
print("records", nrec, "long", nlong, "kmers", int(n.sum()), "len hist", np.bincount(n, minlength=17))
print("d2 hist", np.bincount(meta & 127)[:8], "zero bases", int((bases == 0).sum()))
pw = t._mini_plan[1]
rt = pw[512:512 + 8 * 40].view(torch.int64).cpu().numpy()
print("region totals", rt)
