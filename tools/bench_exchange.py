#!/usr/bin/env python3
"""Rehearse one rank's share of an N-rank step on ONE GPU, at the table geometry the N-rank job would use: the N shards
are generated and counted one after the other (deferred form), their entries are placed in the gather buffer exactly as the
all-gather would leave them, and the table is rebuilt from the N parts -- so the rebuilt table is the true union and the
lookups see its real load.  Everything but the all-gather itself (RCCL over xGMI) is timed."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pangaea_amd import dist as pdist  # noqa: E402
from pangaea_amd import kmer, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=10_000_000)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--load", type=float, default=0.6, help="highest load of the union table")
ap.add_argument("--log2-slots", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
world = args.world


def shard(r):
    cfg = synth.SynthConfig(n_pairs=args.pairs, n_barcodes=max(1, args.pairs // 200), seed=2022, first_pair=r * args.pairs)
    return synth.generate(cfg, device=dev, chunk_pairs=1 << 17, with_names=False)


def timed(msg, f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = f()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"  {msg:58s} {best * 1e3:8.2f} ms", flush=True)
    return out


# size the table as the ranks would: the union's HyperLogLog sketch is the elementwise maximum of the shards' sketches
union = None
for r in range(world):
    regs = kmer.distinct_sketch(shard(r), 21)
    local = kmer.sketch_estimate(regs) if r == 0 else local
    union = regs if union is None else torch.maximum(union, regs)
est = kmer.sketch_estimate(union)
table = (kmer.KmerTable.with_slots(21, dev, args.log2_slots) if args.log2_slots
         else kmer.KmerTable.alloc(21, dev, "hash", distinct_hint=int(1.05 * est), load=args.load))
g = pdist.deferred_group_for(table, int(1.1 * local))
nb = table.n_buckets
print(f"{world} shards x {args.pairs} pairs: ~{local / 1e6:.0f} M distinct per shard, ~{est / 1e6:.0f} M in the union -> 2^{table.log2_slots} slots "
      f"({table.data.numel() * 8 / 1e9:.1f} GB, load {est / table.data.numel():.2f}), {nb} buckets, deferred groups of 2^{g}", flush=True)

fills = torch.zeros((world, nb), dtype=torch.int64, device=dev)
planes = table.tag_bits <= 31            # the 6-byte exchange format (4-byte tags + 2-byte counts)
parts = []
plane_parts = []
overflow = torch.empty(1 << 20, dtype=torch.int64, device=dev)
n_over = torch.zeros(1, dtype=torch.int64, device=dev)
for r in reversed(range(world)):            # shard 0 last: its stream, plan and records stay for the timings below
    s = shard(r)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, dev)
    table.reset().count(s, rows=plan, deferred_group=g)
    fills[r] = table.deferred_fill()
    seg_r = torch.cat([fills.new_zeros(1), torch.cumsum(fills[r], 0)])
    part = torch.empty(int(fills[r].sum()), dtype=torch.int64, device=dev)
    table.deferred_compact_into(part, seg_r)
    parts.append((r, part))
    if planes:
        cap_r = (int(fills[r].sum()) + 7) // 8 * 8
        pl = torch.empty(6 * cap_r, dtype=torch.uint8, device=dev)
        table.deferred_planes_into(pl, seg_r[:-1].contiguous(), (seg_r[:-1] + 2 * cap_r).contiguous(), overflow, n_over)
        plane_parts.append((r, cap_r, pl))
cap = max(p.numel() for _, p in parts)
buf = torch.empty(world * cap, dtype=torch.int64, device=dev)
for r, p in parts:
    buf[r * cap:r * cap + p.numel()] = p
del parts
seg = torch.zeros((world, nb + 1), dtype=torch.int64, device=dev)
seg[:, 1:] = torch.cumsum(fills, 1)
seg0 = seg[0].contiguous()
seg += torch.arange(world, device=dev)[:, None] * cap
print(f"  (all-gather payload: {cap * 8 / 1e9:.2f} GB per rank, {cap * 8 * (world - 1) / 1e9:.2f} GB received)")

mine = torch.empty(cap, dtype=torch.int64, device=dev)
timed(f"K2, deferred (partition + groups of 2^{g} buckets in LDS)", lambda: table.reset().count(s, check=False, rows=plan, deferred_group=g))
table.check_status()
timed("gather of the entries out of the workspace", lambda: table.deferred_compact_into(mine, seg0))
timed(f"rebuild from {world} parts", lambda: table.rebuild_from(buf, seg, check=False), reps=2)
table.check_status()
print(f"  load of the rebuilt table {table.occupancy():.3f}")
if planes:
    want = torch.sort(table.compact()).values
    cap6 = max(c for _, c, _ in plane_parts)
    buf6 = torch.zeros(world * 6 * cap6, dtype=torch.uint8, device=dev)
    for r, cap_r, pl in plane_parts:          # re-lay every part with the common cap (tags | counts)
        buf6[r * 6 * cap6:r * 6 * cap6 + 4 * cap_r] = pl[:4 * cap_r]
        buf6[r * 6 * cap6 + 4 * cap6:r * 6 * cap6 + 4 * cap6 + 2 * cap_r] = pl[4 * cap_r:]
    del plane_parts
    seg6 = (seg - torch.arange(world, device=dev)[:, None] * cap).contiguous()
    mine6 = torch.empty(6 * cap6, dtype=torch.uint8, device=dev)
    table.reset().count(s, check=False, rows=plan, deferred_group=g)
    te, ce = seg6[0, :-1].contiguous(), (seg6[0, :-1] + 2 * cap6).contiguous()
    timed("gather of the entries as 6-byte planes", lambda: table.deferred_planes_into(mine6, te, ce, overflow, n_over.zero_()))
    timed(f"rebuild from {world} parts of 6-byte planes", lambda: table.rebuild_from_planes(buf6, 6 * cap6, cap6, seg6, (0, nb)), reps=2)
    table.check_status()
    print(f"  (6-byte payload: {6 * cap6 / 1e9:.2f} GB per rank, {6 * cap6 * (world - 1) / 1e9:.2f} GB received; overflow entries {int(n_over.item())}); "
          f"same table as from 8-byte entries: {bool(torch.equal(torch.sort(table.compact()).values, want))}")
tnf = torch.zeros((len(rows), kmer.tnf_ncols(4)), dtype=torch.int32, device=dev)
abd = torch.zeros((len(rows), 400), dtype=torch.int32, device=dev)
timed("K1 + K3 (records -> LDS lookups -> row shuffle)", lambda: kmer.features(s, plan, k_tnf=4, table=table, window=10, vsize=400, out_tnf=tnf, out_abd=abd))
_, abd2 = kmer.features(s, rows, k_tnf=None, table=table, window=10, vsize=400)
print(f"  shuffle path == lookup path on the rebuilt table: {bool(torch.equal(abd, abd2))}")
# the non-deferred form, for comparison
timed("K2 writing the rank's own table", lambda: table.reset().count(s, check=False, rows=plan))
timed("  + bucket_fill", table.bucket_fill)
timed("  + compact_into", lambda: table.compact_into(mine, seg0))
