#!/usr/bin/env python3
"""time the LOCAL parts of the multi-GPU table exchange on one GPU, at the geometry an N-rank job would use: the table of
one 10 M-pair shard is compacted, stands in for the N-1 foreign parts as well (an in-device copy replaces the all-gather),
and the table is rebuilt from the N parts.  The all-gather itself (RCCL over xGMI) is not measured here."""
import argparse
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pangaea_amd import kmer, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=10_000_000)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--old", action="store_true", help="also time the torch-ops form (compact/bucket_counts/cat + merge)")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = synth.SynthConfig(n_pairs=args.pairs, n_barcodes=max(1, args.pairs // 200), seed=2022)
s = synth.generate(cfg, device=dev)
distinct = kmer.estimate_distinct(s, 21)
# sized as bench.py sizes the union over ranks (genomic k-mers are shared, error k-mers are not)
table = kmer.KmerTable.alloc(21, dev, "hash", distinct_hint=int(130e6 + 0.7e8 * args.world * args.pairs / 10e6))
log2_slots = table.log2_slots
plan = kmer.Plan(s.rows(2000), dev)
table.count(s, rows=plan)
print(f"{args.pairs} pairs, ~{distinct / 1e6:.0f} M distinct, world {args.world}: 2^{log2_slots} slots ({table.data.numel() * 8 / 1e9:.1f} GB), {table.n_buckets} buckets")


def timed(msg, f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = f()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"  {msg:52s} {best * 1e3:8.2f} ms")
    return out


world, nb = args.world, table.n_buckets
fill = timed("bucket_fill", table.bucket_fill)
fills = fill[None, :].repeat(world, 1)
ends = torch.cumsum(fills, 1)
cap = int(ends[:, -1].max().item())
seg = torch.zeros((world, nb + 1), dtype=torch.int64, device=dev)
seg[:, 1:] = ends
buf = torch.empty(world * cap, dtype=torch.int64, device=dev)
seg0 = seg[0].contiguous()
timed("compact_into (own slot of the gather buffer)", lambda: table.compact_into(buf[:cap], seg0))
for r in range(1, world):
    buf[r * cap:(r + 1) * cap].copy_(buf[:cap])
print(f"  (all-gather payload: {cap * 8 / 1e9:.2f} GB per rank, {cap * 8 * (world - 1) / 1e9:.2f} GB received)")
seg += torch.arange(world, device=dev)[:, None] * cap
timed(f"rebuild_from ({world} parts)", lambda: table.rebuild_from(buf, seg, check=False), reps=2)
table.check_status()
print(f"  occupancy after the rebuild {table.occupancy():.3f}")
if args.old:
    table.reset().count(s, rows=plan)
    comp = timed("torch: compact()", table.compact)
    cnt = timed("torch: bucket_counts()", table.bucket_counts)
    timed(f"torch: merge_parts ({world - 1} foreign parts, cat + merge)", lambda: table.merge_parts([(comp, cnt)] * (world - 1), check=False), reps=1)
