#!/usr/bin/env python3
"""rehearsal of the multi-rank feature step from one FASTQ file: every rank ingests its own byte range, counts, the table
is exchanged, rows are built.  Launch with torch.distributed.run (gloo; all ranks share cuda:0 on a one-GPU box, so the
GPU phases serialise there -- the host phases are what this measures):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 \
        tools/e2e_sharded.py 4000000
"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pangaea_amd import _lib, kmer, synth  # noqa: E402
from pangaea_amd import dist as pdist  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29532")
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
cores = len(os.sched_getaffinity(0))
threads = max(1, min(32, cores // world))
_lib.load().pg_set_ingest_threads(threads)
fq = os.path.join(tempfile.gettempdir(), "pg_e2e_sharded.fq")
if rank == 0:
    cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=max(1, n_pairs // 200), seed=5)
    synth.write_fastq(synth.generate(cfg, device=dev), cfg, fq)
    print(f"{n_pairs} pairs, {os.path.getsize(fq) / 1e6:.0f} MB, {world} ranks x {threads} ingest threads ({cores} cores)", flush=True)
kmer.count_kmers(synth.generate(synth.SynthConfig(n_pairs=2000, n_barcodes=10), device=dev), 21)      # warm the runtime
dist.barrier()


def lap(msg, t):
    torch.cuda.synchronize()
    mine = torch.tensor([time.perf_counter() - t], dtype=torch.float64)
    dist.all_reduce(mine, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(f"  {msg:36s} {1e3 * mine.item():8.1f} ms (slowest rank)", flush=True)
    return time.perf_counter()


t = t_all = time.perf_counter()
host = pdist.ingest_shard(fq); t = lap("ingest own byte range", t)
s = host.to(dev); t = lap("H2D", t)
rows = s.rows(2000); plan = kmer.Plan(rows, dev); t = lap("rows + plan", t)
table = kmer.count_kmers(s, 21, rows=plan); t = lap("K2 table (GPU shared by the ranks)", t)
pdist.exchange_table(table); t = lap("table exchange (gloo, via host)", t)
tnf, abd = kmer.features(s, plan, k_tnf=4, table=table); t = lap("K1 + K3 rows", t)
total = torch.tensor([time.perf_counter() - t_all], dtype=torch.float64)
dist.all_reduce(total, op=dist.ReduceOp.MAX)
pairs = torch.tensor([host.n_pairs]); dist.all_reduce(pairs)
if rank == 0:
    print(f"total {total.item():.2f} s, {int(pairs.item())} pairs over {world} ranks", flush=True)
    os.remove(fq)
dist.destroy_process_group()
