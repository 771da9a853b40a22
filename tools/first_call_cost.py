import sys, time, torch
sys.path.insert(0, "/root/repo")
from pangaea_amd import kmer, synth
def lap(msg, t):
    torch.cuda.synchronize(); now=time.perf_counter(); print(f"{msg:40s} {1e3*(now-t):8.1f} ms", flush=True); return now
dev="cuda:0"
kmer.count_kmers(synth.generate(synth.SynthConfig(n_pairs=2000, n_barcodes=10), device=dev), 21)
s = synth.generate(synth.SynthConfig(n_pairs=4_000_000, n_barcodes=20_000, seed=5), device=dev, chunk_pairs=1<<17, with_names=False)
rows = s.rows(2000); plan = kmer.Plan(rows, dev)
t=time.perf_counter()
d = kmer.estimate_distinct(s, 21); t=lap("HLL", t)
table = kmer.KmerTable.alloc(21, dev, "hash", int(1.1*d), load=0.4); t=lap("alloc table", t)
ws = table._workspace_for(s.n_words); t=lap("alloc count workspace", t)
sws = table._shuffle_workspace_for(s.n_words, plan.n_rows, 400); t=lap("alloc shuffle workspace", t)
table.count(s, rows=plan, emit=(10,400)); t=lap("count (first)", t)
tnf, abd = kmer.features(s, plan, k_tnf=4, table=table); t=lap("features (first)", t)
table.reset().count(s, rows=plan, emit=(10,400)); t=lap("count (second)", t)
tnf, abd = kmer.features(s, plan, k_tnf=4, table=table); t=lap("features (second)", t)
