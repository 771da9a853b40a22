"""Multi-GPU decomposition of the feature path (one process per GPU, torch.distributed; "nccl" is RCCL on ROCm).

The reference is single-process; the decomposition is the build's own:
  * rows (barcode runs) are independent -> every rank takes a contiguous range of runs and produces its own block of
    the count matrices: no data-path collective for K1/K3 or the VAE encode.  An uncompressed interleaved file is
    already cut at ingest (byte ranges moved to run boundaries, `ingest_shard`), so no rank ever parses or holds
    more than its share; other inputs are parsed whole and cut by runs, balanced by characters (`shard_stream`);
  * the k-mer multiplicity table is a global sum -> ONE exchange after counting: dense tables (k <= 16) are
    summed with an all-reduce; hash tables are compacted bucket by bucket straight into one gather buffer,
    all-gathered in place, and rebuilt from all parts -- one workgroup per bucket inside LDS, because every rank uses
    the same bucket geometry -- after which every rank holds the full table and looks up locally.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

from .kmer import KmerTable
from .reads import ReadStream


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


# ---- the control plane.  A rank that waits for ANOTHER rank's host-side work (rank 0 training the VAE, writing the cache
# files, running the assemblers) must not wait inside a collective of the data group: RCCL's watchdog aborts the process
# once a collective has been pending for the group's timeout (10 minutes by default), and torchrun then tears the working
# rank down with it.  Such waits -- and the tiny host-side decisions next to them -- go over a gloo group of their own whose
# timeout is a month; the data group only ever carries collectives that every rank enters at about the same time.
_CONTROL = None
CONTROL_TIMEOUT_DAYS = 30


def control_group():
    """the gloo control group over all ranks (created on first use: EVERY rank must make that first call at the same point of
    the program, which ``pangaea.run`` does right after ``init_process_group``); None outside a multi-rank run"""
    global _CONTROL
    if not is_distributed():
        return None
    if _CONTROL is None:
        import datetime
        _CONTROL = dist.new_group(backend="gloo", timeout=datetime.timedelta(days=CONTROL_TIMEOUT_DAYS))
    return _CONTROL


def agreed(value, src: int = 0):
    """rank ``src``'s value on every rank (control plane: may be waited for as long as ``src`` takes to get here)"""
    if not is_distributed():
        return value
    box = [value]
    dist.broadcast_object_list(box, src=src, group=control_group())
    return box[0]


def wait_for_all() -> None:
    """a barrier on the control plane: safe to sit in while another rank does minutes or hours of host work"""
    if is_distributed():
        dist.barrier(group=control_group())


def leave() -> None:
    """behind the LAST collective of a run: every rank meets once on the control plane and the process groups are destroyed.
    Ranks without further work return to their caller and exit; the one that goes on (file writing, bin extraction, the
    assembly stage) does so without a group -- nobody is left waiting in a collective for it."""
    global _CONTROL
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1:
            dist.barrier(group=control_group())
        _CONTROL = None
        dist.destroy_process_group()


def balanced_run_ranges(run_off: np.ndarray, world: int) -> list[tuple[int, int]]:
    """contiguous run ranges [(first, last+1)] per rank with roughly equal character counts; ranges tile all runs"""
    n = len(run_off) - 1
    total = int(run_off[-1] - run_off[0])
    cuts = [0]
    for r in range(1, world):
        target = run_off[0] + total * r // world
        i = int(np.searchsorted(run_off, target, side="left"))
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_stream(stream: ReadStream, rank: int, world: int) -> ReadStream:
    """the part of a (host) stream a rank counts and builds rows for: its run range, word aligned.

    Runs keep their reference shape (own pairs 2..n + first pair of the next barcode) because sharding happens
    AFTER run assembly: the one-pair halo of SURVEY 8e is already inside the run.  Characters outside every run
    (reads of skipped pairs, stored after the last run) go to the last rank; they only feed the global table.
    """
    first, last = balanced_run_ranges(stream.run_off, world)[rank]
    c0 = int(stream.run_off[first])
    c1 = int(stream.run_off[last]) if rank < world - 1 else stream.n_chars
    w0, w1 = c0 // 32, (c1 + 31) // 32
    codes = stream.codes[w0:w1].clone()
    valid = stream.valid[w0:w1].clone()
    # characters of neighbouring ranks that share the first / last word are masked out (they are counted there)
    if codes.numel():
        lo, hi = c0 - 32 * w0, c1 - 32 * (w1 - 1)
        keep_first = ~((1 << lo) - 1) & 0xFFFFFFFF
        keep_last = ((1 << hi) - 1) & 0xFFFFFFFF if hi < 32 else 0xFFFFFFFF
        v = valid.to(torch.int64) & 0xFFFFFFFF
        v[0] &= keep_first
        v[-1] &= keep_last
        valid = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)
    from .reads import words_for
    pad = words_for((w1 - w0) * 32) - (w1 - w0)
    if pad:
        codes = torch.cat([codes, codes.new_zeros(pad)])
        valid = torch.cat([valid, valid.new_zeros(pad)])
    run_off = stream.run_off[first:last + 1] - 32 * w0
    lower = None
    if stream.valid_lower is not None:          # the lower-case plane is cut and masked exactly like `valid`
        lower = stream.valid_lower[w0:w1].clone()
        if lower.numel():
            lv = lower.to(torch.int64) & 0xFFFFFFFF
            lv[0] &= keep_first
            lv[-1] &= keep_last
            lower = torch.where(lv >= (1 << 31), lv - (1 << 32), lv).to(torch.int32)
        if pad:
            lower = torch.cat([lower, lower.new_zeros(pad)])
    lowq = None
    if stream.valid_lowq is not None:           # the quality plane too
        lowq = stream.valid_lowq[w0:w1].clone()
        if lowq.numel():
            qv = lowq.to(torch.int64) & 0xFFFFFFFF
            qv[0] &= keep_first
            qv[-1] &= keep_last
            lowq = torch.where(qv >= (1 << 31), qv - (1 << 32), qv).to(torch.int32)
        if pad:
            lowq = torch.cat([lowq, lowq.new_zeros(pad)])
    return ReadStream(codes, valid, c1 - 32 * w0, run_off.astype(np.int64), stream.run_names[first:last], mode=stream.mode,
                      valid_lower=lower, valid_lowq=lowq)


def ingest_shard(reads1: str, reads2: str | None = None, group=None, device="cpu") -> ReadStream:
    """this rank's part of the input (a host stream unless ``device`` is a GPU and the byte-range ingest applies, which then
    copies its pieces there while it parses: ``pg_ingest_fastq_device``).

    An uncompressed interleaved file (what ``run_pangaea`` sorts the reads into) is cut by bytes: every rank counts the
    newlines of its own range, the counts are all-gathered (the only communication), and each rank reads and parses
    just its range, moved to run boundaries (``pg_ingest_fastq_shard``) -- host work and memory per rank fall with the
    number of ranks.  gzip and -1/-2 inputs cannot be cut by bytes: there every rank parses the file and keeps its
    balanced range of runs (``shard_stream``)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sharded = reads2 is None
    if sharded:
        with open(reads1, "rb") as f:
            sharded = f.read(2) != b"\x1f\x8b"
    if not sharded:
        return shard_stream(ReadStream.from_fastq(reads1, reads2), rank, world)
    mine = torch.tensor([ReadStream.count_newlines(reads1, rank, world)], dtype=torch.int64)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        counts = [c.to(dev) for c in counts]
        dist.all_gather(counts, mine.to(dev), group=group)
    else:
        dist.all_gather(counts, mine, group=group)
    before = np.concatenate([[0], np.cumsum([int(c.item()) for c in counts])]).astype(np.int64)
    return ReadStream.from_fastq_shard(reads1, rank, world, before, device=device)


def _staged(t: torch.Tensor, group=None) -> torch.Tensor:
    """gloo cannot all-gather device tensors: stage through the host there (RCCL takes them as they are)"""
    return t.cpu() if t.is_cuda and dist.get_backend(group) == "gloo" else t


EXCHANGE_RANGES = 4          # bucket ranges of the pipelined table exchange


def _all_gather_flat(out: torch.Tensor, mine: torch.Tensor, group=None, async_op: bool = False):
    """all-gather of equal-sized contiguous vectors into one flat buffer (RCCL: one collective, no list copies; with
    ``async_op`` the work handle is returned and the caller's stream waits on it only when told to);
    gloo: staged through the host, synchronously"""
    if dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(out, mine, group=group, async_op=async_op) if async_op else \
            dist.all_gather_into_tensor(out, mine, group=group)
    n = mine.numel()
    host = [torch.empty(n, dtype=out.dtype) for _ in range(dist.get_world_size(group))]
    dist.all_gather(host, mine.cpu().contiguous(), group=group)
    for r, h in enumerate(host):
        out[r * n:(r + 1) * n].copy_(h)
    return None


def gather_pairs(local: torch.Tensor, group=None) -> list[torch.Tensor]:
    """all-gather of variable-length int64 vectors (padded to the longest); returns one tensor per rank"""
    world = dist.get_world_size(group)
    home = local.device
    local = _staged(local, group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = local.new_zeros(cap)
    buf[:local.numel()] = local
    out = [local.new_zeros(cap) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return [o[:s].to(home) for o, s in zip(out, sizes)]


def everyone(flag: bool, group=None) -> bool:
    """the same answer on every rank: did ALL ranks say yes?  (decisions that select between collectives -- or between doing
    one and not -- must not be taken rank by rank).  ``group=None``: the control plane when there is one, so that the
    question may be asked right behind a long stretch of one rank's host work."""
    if group is None and _CONTROL is not None:
        group = _CONTROL
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        t = t.to(torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


def gather_rows(local: torch.Tensor, dst: int = 0, group=None) -> torch.Tensor | None:
    """the ranks' row blocks [n_r, D] (any dtype, D equal) stacked in rank order ON RANK ``dst`` only (None elsewhere):
    how the row-sharded matrices reach the one rank that writes the cache files -- nothing is replicated, nothing is
    widened (rows travel as they are, padded to the longest block)"""
    world, me = dist.get_world_size(group), dist.get_rank(group)
    home = local.device
    staged = _staged(local.contiguous(), group)
    n = torch.tensor([staged.shape[0]], dtype=torch.int64, device=staged.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(v.item()) for v in sizes]
    cap = max(max(sizes), 1)
    buf = staged.new_zeros((cap,) + tuple(staged.shape[1:]))
    buf[:staged.shape[0]] = staged
    out = [torch.empty_like(buf) for _ in range(world)] if me == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if me != dst:
        return None
    return torch.cat([o[:n_r] for o, n_r in zip(out, sizes)]).to(home)


def deferred_group_for(table: KmerTable, local_distinct: int) -> int | None:
    """g for ``table.count(deferred_group=g)``: how many times sparser this rank's keys are than the union the table is
    sized for (as a power of two), or None where the deferred form does not apply"""
    if not (table.kind == "hash" and table._bucketed() and table.log2_slots - table.log2_bucket > 8):
        return None
    local_log2 = max(1, (int(local_distinct / 0.45) - 1).bit_length())        # slots this rank alone would need
    return max(0, table.log2_slots - local_log2)


def count_kmers_sharded(stream: ReadStream, k: int, rows=None, group=None, max_log2_slots: int = 36,
                        lowercase_is_base: bool = False) -> KmerTable:
    """the global table of all ranks' streams, on every rank: the distributed form of ``kmer.count_kmers``.

    The ranks' HyperLogLog sketches are combined with an all-reduce(MAX) -- the sketch of the union -- so every rank
    allocates the same table, sized for the union; hash tables are then counted in deferred form where that applies
    (the rank's own sparse table is never written), exchanged once and rebuilt.  A full table (every rank sees the same
    status after the rebuild) is re-done with four times the slots on all ranks."""
    from . import _lib, kmer
    kind = KmerTable.default_kind(k)
    if kind == "dense":
        return exchange_table(KmerTable.alloc(k, stream.device, kind).count(stream, lowercase_is_base=lowercase_is_base), group)
    regs = kmer.distinct_sketch(stream, k, lowercase_is_base=lowercase_is_base)
    local = kmer.sketch_estimate(regs)
    union = _staged(regs.clone(), group)
    dist.all_reduce(union, op=dist.ReduceOp.MAX, group=group)
    total = max(1 << 13, int(1.05 * kmer.sketch_estimate(union)))
    table = KmerTable.alloc(k, stream.device, kind, total, load=0.6)      # measured: fewer, fuller buckets beat a sparser table
    def any_full() -> bool:
        # the same answer on every rank, so that all regrow together (a group table in LDS can be full on one rank only);
        # nothing between two collectives may raise on one rank alone
        flag = ((table.status[:1] & 1) != 0).to(torch.int32)         # (bit 1 = overflow list of the exchange: handled there)
        flag = _staged(flag, group)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        return int(flag.item()) != 0

    def everyone(flag: bool) -> bool:
        t = _staged(torch.tensor([1 if flag else 0], dtype=torch.int32, device=stream.device), group)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(t.item())

    while True:
        g = deferred_group_for(table, int(1.1 * local)) if kind == "hash" else None
        # the exchange takes different collectives for deferred and materialised counts: all ranks must take the same form
        if everyone(g is not None and table.can_defer(stream.n_words)):
            table.count(stream, rows=rows, deferred_group=g, check=False, lowercase_is_base=lowercase_is_base)
        else:
            table.count(stream, rows=rows, check=False, lowercase_is_base=lowercase_is_base)
        full = any_full()
        if not full:
            exchange_table(table, group, check=False)
            full = any_full()
        if not full:
            return table
        if table.log2_slots >= max_log2_slots:
            raise _lib.PangaeaError(_lib.PG_ETABLEFULL, f"hash table with 2^{table.log2_slots} slots is full")
        log2 = min(max_log2_slots, table.log2_slots + 2)
        wide = table.kind == "wide"
        del table
        table = KmerTable.wide_with_slots(k, stream.device, log2) if wide else KmerTable.with_slots(k, stream.device, log2)


PLANES = True                # send deferred entries as 4-byte tags + 2-byte counts (6 B instead of 8 B per entry)


def _exchange_planes(table: KmerTable, group, me: int, world: int, seg, cuts, at, sizes) -> None:
    """the range-wise exchange in the 6-byte format: per bucket range every rank sends [tags uint32 x cap | counts uint16 x
    cap] (cap = the longest part of that range, rounded to 8); counts >= 0xffff send their remainder in a small overflow
    list of whole entries, merged at the end -- so the rebuilt counts are exact"""
    dev = seg.device
    n_ranges = len(cuts) - 1
    nb = table.n_buckets
    caps = [max(8, (int(c) + 7) // 8 * 8) for c in sizes.max(dim=0).values.cpu().tolist()]          # host sync: buffer sizes
    stride = [6 * c for c in caps]                                   # bytes per rank and range (a multiple of 16)
    base = [0]
    for c in range(n_ranges):
        base.append(base[-1] + stride[c])
    # where bucket b's first tag / count goes, as elements of the uint32 / uint16 views of this rank's send buffer
    which = torch.bucketize(torch.arange(nb, device=dev), torch.tensor(cuts[1:-1], device=dev, dtype=torch.int64), right=True)
    j0 = seg[me, :-1] - at[me][which]
    base_t = torch.tensor(base[:-1], device=dev, dtype=torch.int64)
    caps_t = torch.tensor(caps, device=dev, dtype=torch.int64)
    tag_elem = (base_t[which] // 4 + j0).contiguous()
    cnt_elem = ((base_t[which] + 4 * caps_t[which]) // 2 + j0).contiguous()
    mine = torch.empty(base[-1], dtype=torch.uint8, device=dev)
    overflow = torch.empty(max(1 << 16, int(table.deferred_fill().numel())), dtype=torch.int64, device=dev)
    n_over = torch.zeros(1, dtype=torch.int64, device=dev)
    table.deferred_planes_into(mine, tag_elem, cnt_elem, overflow, n_over)
    bufs, works = [], []
    for c in range(n_ranges):
        buf = torch.empty(world * stride[c], dtype=torch.uint8, device=dev)
        works.append(_all_gather_flat(buf, mine[base[c]:base[c + 1]], group, async_op=True))
        bufs.append(buf)
    for c in range(n_ranges):
        seg_c = (seg[:, cuts[c]:cuts[c + 1] + 1] - at[:, c:c + 1]).contiguous()
        if works[c] is not None:
            works[c].wait()
        table.rebuild_from_planes(bufs[c], stride[c], caps[c], seg_c, (cuts[c], cuts[c + 1]))
    # counts beyond 0xffff: their remainders, from every rank (this one included), as whole entries
    for part in _gather_overflow(overflow, n_over, group):
        if part.numel():
            table.merge(part, check=False)


OWNER_MIN_WORLD = 4          # from this many ranks on, partial tables are reduced at bucket-range owners before they are replicated


def _all_to_all_flat(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    """all-to-all of equal byte splits (the r-th slice of ``inp`` goes to rank r, into slot ``me`` of its ``out``)"""
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(out, inp, group=group)
        return
    host = torch.empty(out.numel(), dtype=out.dtype)
    dist.all_to_all_single(host, inp.cpu().contiguous(), group=group)
    out.copy_(host)


def _gather_overflow(overflow: torch.Tensor, n_over: torch.Tensor, group) -> list[torch.Tensor]:
    """the ranks' overflow lists (remainders of counts >= 0xffff).  A list that ran over its capacity is an error -- raised on
    EVERY rank and only after the collective, so no rank is left waiting in it"""
    n = int(n_over.item())
    parts = gather_pairs(overflow[:min(n, overflow.numel())].contiguous(), group)
    if not everyone(n <= overflow.numel(), group):
        raise RuntimeError(f"exchange overflow list too small (this rank: {n} entries for {overflow.numel()} places)")
    return parts


def _exchange_owner(table: KmerTable, group, me: int, world: int, seg) -> None:
    """reduce-scatter + all-gather form of the exchange (6-byte entries throughout).

    An all-gather of the partial tables sends every shared k-mer world - 1 times to every rank.  Here rank o OWNS bucket
    range o: (1) all-to-all -- every rank sends its entries of range o to o; (2) o rebuilds its range from the world parts
    inside LDS; (3) the merged ranges, which hold every k-mer once, are all-gathered and every rank rebuilds the ranges it
    does not own from that single part.  At 8 ranks x 10 M pairs a rank receives 0.9 + 2.8 GB instead of 7.4 GB."""
    from . import _lib
    dev = seg.device
    nb = table.n_buckets
    cuts = [nb * o // world for o in range(world + 1)]
    at = seg[:, cuts]                                                            # [part, world + 1]
    sizes = at[:, 1:] - at[:, :-1]                                               # [part, owner]
    cap1 = max(8, (int(sizes.max().item()) + 7) // 8 * 8)                        # host sync: buffer size
    s1 = 6 * cap1
    which = torch.bucketize(torch.arange(nb, device=dev), torch.tensor(cuts[1:-1], device=dev, dtype=torch.int64), right=True)
    j0 = seg[me, :-1] - at[me][which]
    tag_elem = (which * (s1 // 4) + j0).contiguous()
    cnt_elem = (which * (s1 // 2) + 2 * cap1 + j0).contiguous()
    send = torch.empty(world * s1, dtype=torch.uint8, device=dev)
    overflow = torch.empty(max(1 << 16, nb), dtype=torch.int64, device=dev)
    n_over = torch.zeros(1, dtype=torch.int64, device=dev)
    table.deferred_planes_into(send, tag_elem, cnt_elem, overflow, n_over)
    recv = torch.empty(world * s1, dtype=torch.uint8, device=dev)
    _all_to_all_flat(recv, send, group)
    mine = (cuts[me], cuts[me + 1])
    seg_me = (seg[:, mine[0]:mine[1] + 1] - at[:, me:me + 1]).contiguous()
    table.rebuild_from_planes(recv, s1, cap1, seg_me, mine, in_order=False)
    shift = _lib.HASH_COUNT_BITS + table.tag_bits                               # slot >> shift = bucket id
    for part in _gather_overflow(overflow, n_over, group):                      # remainders of counts >= 0xffff that belong here
        if part.numel():
            b = (part >> shift) & (nb - 1)                                      # (arithmetic shift of an int64: mask the sign bits)
            own = part[(b >= mine[0]) & (b < mine[1])]
            if own.numel():
                table.merge(own.contiguous(), check=False, pending_ok=True)
    # (3) the merged range goes to everybody
    n_mine = mine[1] - mine[0]
    longest = max(cuts[o + 1] - cuts[o] for o in range(world))
    fill2 = torch.zeros(longest, dtype=torch.int64, device=dev)
    fill2[:n_mine] = table.bucket_fill_range(mine)
    fills2 = torch.empty((world, longest), dtype=torch.int64, device=dev)
    _all_gather_flat(fills2.view(-1), fill2, group)
    ends2 = torch.cumsum(fills2, dim=1)
    seg2 = torch.zeros((world, longest + 1), dtype=torch.int64, device=dev)
    seg2[:, 1:] = ends2
    # the merged ranges travel in EXCHANGE_RANGES chunks (sub-ranges of every owner's range), all issued at once on RCCL's
    # stream: the rebuild of one chunk runs while the next is still in flight
    n_chunks = EXCHANGE_RANGES if longest >= 64 * EXCHANGE_RANGES else 1
    local = [[(cuts[o + 1] - cuts[o]) * c // n_chunks for c in range(n_chunks + 1)] for o in range(world)]      # per owner, in its own buckets
    at2 = torch.stack([seg2[o, local[o]] for o in range(world)])                 # [owner, n_chunks + 1]
    caps2 = [max(8, (int(v) + 7) // 8 * 8) for v in (at2[:, 1:] - at2[:, :-1]).max(dim=0).values.cpu().tolist()]   # host sync: buffer sizes
    stride2 = [6 * c for c in caps2]
    base2 = [0]
    for c in range(n_chunks):
        base2.append(base2[-1] + stride2[c])
    which2 = torch.bucketize(torch.arange(n_mine, device=dev), torch.tensor(local[me][1:-1], device=dev, dtype=torch.int64), right=True)
    j2 = seg2[me, :n_mine] - at2[me][which2]
    base2_t = torch.tensor(base2[:-1], device=dev, dtype=torch.int64)
    caps2_t = torch.tensor(caps2, device=dev, dtype=torch.int64)
    send2 = torch.empty(base2[-1], dtype=torch.uint8, device=dev)
    n_over.zero_()
    table.compact_planes_range(mine, send2, (base2_t[which2] // 4 + j2).contiguous(),
                               ((base2_t[which2] + 4 * caps2_t[which2]) // 2 + j2).contiguous(), overflow, n_over)
    bufs, works = [], []
    for c in range(n_chunks):
        buf = torch.empty(world * stride2[c], dtype=torch.uint8, device=dev)
        works.append(_all_gather_flat(buf, send2[base2[c]:base2[c + 1]], group, async_op=True))
        bufs.append(buf)
    for c in range(n_chunks):
        if works[c] is not None:
            works[c].wait()
        for o in range(world):
            lo, hi = local[o][c], local[o][c + 1]
            if o != me and hi > lo:
                seg_o = (seg2[o:o + 1, lo:hi + 1] - at2[o, c]).contiguous()
                table.rebuild_from_planes(bufs[c][o * stride2[c]:(o + 1) * stride2[c]], stride2[c], caps2[c], seg_o,
                                          (cuts[o] + lo, cuts[o] + hi), in_order=False)
    table.mark_rebuilt()
    for r, part in enumerate(_gather_overflow(overflow, n_over, group)):        # the owners' remainders: ranges this rank does not own
        if r != me and part.numel():
            table.merge(part, check=False)


def _exchange_bucketed(table: KmerTable, group=None) -> None:
    """the exchange of a bucketed hash table (also callable in a one-rank group, which a one-GPU box can hold over RCCL)"""
    me = dist.get_rank(group)
    world = dist.get_world_size(group)
    # Every rank built its table with the same geometry.  Per-bucket fills are exchanged first (8 B per bucket); they give
    # the segment offsets of every rank's bucket-ordered compaction.  The compactions are gathered (padded to the longest
    # part; the padding is never read) and the table is rebuilt bucket by bucket inside LDS from all parts, the own one
    # included -- the sparse slices are read once (to compact) and written once (the merged image); nothing is
    # concatenated or scanned by torch.  A deferred count (KmerTable.count(deferred_group=g)) has its fills and entries
    # already and never wrote its sparse table: the compaction is then a gather out of the count's workspace.
    fill = table.deferred_fill() if table.pending else table.bucket_fill()
    nb = table.n_buckets
    dev = fill.device
    fills = torch.empty((world, nb), dtype=torch.int64, device=dev)
    _all_gather_flat(fills.view(-1), fill, group)
    seg = torch.zeros((world, nb + 1), dtype=torch.int64, device=dev)          # every part's offsets, from its own start
    seg[:, 1:] = torch.cumsum(fills, dim=1)
    # the gather is cut into bucket ranges so that the rebuild of one range runs while the next is still in flight
    # (the collectives queue on RCCL's stream, the rebuilds on the compute stream, each waiting for its own range only)
    n_ranges = EXCHANGE_RANGES if nb >= 64 * EXCHANGE_RANGES else 1
    cuts = [nb * c // n_ranges for c in range(n_ranges + 1)]
    at = seg[:, cuts]                                                            # [world, n_ranges + 1]
    sizes = at[:, 1:] - at[:, :-1]
    form = os.environ.get("PG_EXCHANGE", "")          # "" = choose; "allgather8" / "allgather6" / "owner" force one form (debugging)
    if table.pending and table.tag_bits <= 31 and PLANES and form != "allgather8":
        if form == "owner" or (form == "" and world >= OWNER_MIN_WORLD and nb >= 64 * world):
            _exchange_owner(table, group, me, world, seg)
        else:
            _exchange_planes(table, group, me, world, seg, cuts, at, sizes)
        return
    host = torch.cat([sizes.max(dim=0).values, at[me]]).cpu().tolist()          # the step's one host sync
    caps, mine_at = [max(int(c), 1) for c in host[:n_ranges]], host[n_ranges:]
    mine = torch.empty(int(mine_at[-1]) + max(caps), dtype=torch.int64, device=dev)   # slack: a range is sent padded
    (table.deferred_compact_into if table.pending else table.compact_into)(mine, seg[me].contiguous())
    bufs, works = [], []
    for c in range(n_ranges):
        buf = torch.empty(world * caps[c], dtype=torch.int64, device=dev)
        works.append(_all_gather_flat(buf, mine[int(mine_at[c]):int(mine_at[c]) + caps[c]], group, async_op=True))
        bufs.append(buf)
    lanes = torch.arange(world, device=dev, dtype=torch.int64)[:, None]
    for c in range(n_ranges):
        seg_c = (seg[:, cuts[c]:cuts[c + 1] + 1] - at[:, c:c + 1] + lanes * caps[c]).contiguous()
        if works[c] is not None:
            works[c].wait()
        table.rebuild_from(bufs[c], seg_c, check=False, buckets=(cuts[c], cuts[c + 1]))


def exchange_table(table: KmerTable, group=None, check: bool = True) -> KmerTable:
    """turn per-rank partial tables into the global table on every rank (the path's only collective)"""
    if not is_distributed():
        if table.pending:
            raise RuntimeError("a deferred count needs a process group to exchange with")
        return table
    if table.kind == "dense":
        data = _staged(table.data, group)
        dist.all_reduce(data, op=dist.ReduceOp.SUM, group=group)           # int32 sum == uint32 sum (mod 2^32)
        if data is not table.data:
            table.data.copy_(data)
        return table
    me = dist.get_rank(group)
    world = dist.get_world_size(group)
    if table.kind == "wide":
        # wide tables (k > 21): gather occupied keys and their counts, add the other ranks' entries with global atomics
        import ctypes as C
        from . import _lib
        from .kmer import _stream_ptr
        keys, cnts = table._wide_parts()
        occ = keys != 0
        all_codes = gather_pairs((keys[occ] - 1).contiguous(), group)
        all_counts = gather_pairs(cnts[occ].to(torch.int64).contiguous(), group)
        for r in range(world):
            if r != me and all_codes[r].numel():
                c, n = all_codes[r].contiguous(), all_counts[r].to(torch.int32).contiguous()
                with torch.cuda.device(table.device):
                    _lib.check(_lib.load().pg_kmer_merge_wide(c.data_ptr(), n.data_ptr(), c.numel(), table.desc(),
                                                              table.status.data_ptr(), _stream_ptr(table.device)))
        if check:
            table.check_status()
        return table
    if table._bucketed():
        _exchange_bucketed(table, group)
    else:
        parts = gather_pairs(table.compact(), group)
        for r, pairs in enumerate(parts):
            if r != me and pairs.numel():
                table.merge(pairs, check=False)
    if check:
        table.check_status()
    return table


# ------------------------------------------------------------------ the super-k-mer form on N > 1 ranks
#
# Round 2's multi-rank path ran round 1's key-partitioned pipeline on every rank (one 8-byte record per k-mer occurrence through
# two scatter passes), replicated the merged table on every rank and probed it once per occurrence: 70 ms of local work per
# 10 M pairs against 34 ms on one GPU.  Here every rank runs the one-GPU pipeline on its own reads up to the provisional
# (row, local slot) words; what travels is one 8-byte entry per DISTINCT k-mer of the rank to the bucket's owner and a 2-byte
# bin back.  The owners hold the global table (each its bucket range); nobody rebuilds or re-probes anything.


class MiniSharded:
    """global multiplicities and abundance rows of one rank's reads, with the other ranks' reads counted in.

    ``union`` is the global table (kind "mini", the union's geometry; this rank writes -- and afterwards holds -- the slices of
    the buckets it owns), ``local`` the rank's own counting geometry: the same buckets, slots for its own k-mers."""

    MIN_LOG2_BUCKETS = 9                 # (two scatter passes; the lookup half reads the per-bucket word totals of the second)

    def __init__(self, k: int, device, union_log2_slots: int, local_log2_bucket: int, window: int, vsize: int, group=None,
                 union_log2_bucket: int | None = None):
        from . import _lib
        lb_u = min(_lib.BUCKET_MAX_LOG2_SLOTS, union_log2_slots) if union_log2_bucket is None else union_log2_bucket
        bits = union_log2_slots - lb_u
        if not (self.MIN_LOG2_BUCKETS <= bits <= _lib.MINI_MAX_LOG2_BUCKETS) or not _lib.MINI_MIN_K <= k <= _lib.HASH_MAX_K:
            raise ValueError(f"MiniSharded needs 2^{self.MIN_LOG2_BUCKETS}..2^{_lib.MINI_MAX_LOG2_BUCKETS} buckets and {_lib.MINI_MIN_K} <= k <= {_lib.HASH_MAX_K}")
        if not 4 <= local_log2_bucket <= lb_u:
            raise ValueError("local buckets hold between 2^4 slots and the union's")
        self.group, self.window, self.vsize = group, int(window), int(vsize)
        self.union = KmerTable.mini_with_slots(k, device, union_log2_slots, lb_u)
        # (the local table object carries geometry, plan and workspaces; its slots are never written: one word stands in)
        self.local = KmerTable(k, "mini", torch.zeros(1, dtype=torch.int64, device=device), bits + local_log2_bucket, local_log2_bucket)
        self.bytes_sent = self.bytes_received = 0

    @staticmethod
    def geometry(union_distinct: int, local_distinct: int, union_load: float = 0.6, local_load: float = 0.45):
        """(union_log2_slots, union_log2_bucket, local_log2_bucket) for these HyperLogLog estimates"""
        from . import _lib
        import math
        log2_u = max(MiniSharded.MIN_LOG2_BUCKETS + 4, math.ceil(math.log2(max(1024.0, union_distinct / union_load))))
        lb_u = min(_lib.BUCKET_MAX_LOG2_SLOTS, log2_u - MiniSharded.MIN_LOG2_BUCKETS)
        bits = log2_u - lb_u
        need = max(16.0, local_distinct / local_load / (1 << bits))
        lb_l = min(lb_u, max(4, math.ceil(math.log2(need))))
        return log2_u, lb_u, lb_l

    def count(self, stream: ReadStream, plan, check: bool = True) -> "MiniSharded":
        """count half on this rank, entries to the owners, merged bins back, lookup half: afterwards ``kmer.features(stream, plan,
        table=self.local, window, vsize)`` reads the abundance rows from the shuffled words"""
        self.count_half(stream, plan)
        self.exchange()
        self.lookup_half()
        if check:
            bits = self.status_bits()
            if bits & self._overflow_bit():
                # the parts of this batch did not fit the size kept from an earlier one (``exchange``): ask again, count again
                self._cap1 = None
                self.local.status.zero_(); self.union.status.zero_()
                return self.count(stream, plan, check=True)
            self.check_status(bits)
        return self

    @staticmethod
    def _overflow_bit() -> int:
        from . import _lib
        return _lib.STATUS_OVERFLOW_LIST

    def count_half(self, stream: ReadStream, plan) -> None:
        """plan -> first and second scatter pass -> the bucket workgroups' count half (this rank's reads only)"""
        self.local.reset(); self.union.reset()
        self.local.count_half(stream, plan, (self.window, self.vsize), check=False)

    def exchange(self) -> None:
        """entries -> owners (all-to-all, 8 bytes per distinct k-mer of this rank) -> merged inside LDS by the owners, which keep the
        slices of the global table -> the bins of exactly the entries received, in order, back (all-to-all, 2 bytes each).

        The collectives move equal parts, padded to ``cap1`` entries per (sender, owner).  The FIRST exchange of an object reads
        the longest part from the device (a host wait) and keeps it with 3 % of slack; every later one reuses that size without
        asking -- no host wait inside a step of a stream of batches.  A batch whose longest part does not fit is not exchanged at
        all (the fills are zeroed on the device, PG_STATUS_OVERFLOW_LIST is raised in the status word, the lookup half returns at
        once): ``count``/``check_status`` see the bit, forget the size and the count is done again."""
        from . import _lib
        group = self.group
        world, me = dist.get_world_size(group), dist.get_rank(group)
        loc, uni = self.local, self.union
        fill = loc._half[0]
        nb = loc.n_buckets
        dev = fill.device
        L = _lib.load()
        cuts = [nb * o // world for o in range(world + 1)]
        const = getattr(self, "_exchange_const", None)
        if const is None or const[0] != (nb, world, str(dev)):
            # (what depends on the geometry alone: built once, not copied to the device in every step)
            cut_idx = torch.tensor(cuts, device=dev, dtype=torch.int64)
            which = torch.bucketize(torch.arange(nb, device=dev), cut_idx[1:-1].contiguous(), right=True)
            const = self._exchange_const = ((nb, world, str(dev)), cut_idx, which)
        _, cut_idx, which = const
        fills = torch.empty((world, nb), dtype=torch.int64, device=dev)
        _all_gather_flat(fills.view(-1), fill, group)

        def parts(f):
            seg = torch.zeros((world, nb + 1), dtype=torch.int64, device=dev)
            seg[:, 1:] = torch.cumsum(f, dim=1)
            at = seg.index_select(1, cut_idx)                                        # [part, world + 1]
            return seg, at, at[:, 1:] - at[:, :-1]                                   # sizes: [part, owner]

        seg, at, sizes = parts(fills)
        cap1 = getattr(self, "_cap1", None)
        if cap1 is None:
            longest = int(sizes.max().item())                                        # host wait: once per object (or after an overflow)
            cap1 = self._cap1 = max(8, (longest + longest // 32 + 7) // 8 * 8)
        else:
            # the same for every rank (the fills are all-gathered): too long a part -> nothing is exchanged, everybody raises the bit
            fits = (sizes.max() <= cap1)
            loc.status[0] |= (~fits).to(loc.status.dtype) * _lib.STATUS_OVERFLOW_LIST
            fills = fills * fits.to(fills.dtype)
            fill = fill * fits.to(fill.dtype)
            seg, at, sizes = parts(fills)
        elem = (which * cap1 + seg[me, :-1] - at[me].index_select(0, which)).contiguous()   # where bucket b's entries (and, later, bins) lie
        send = torch.empty(world * cap1, dtype=torch.int64, device=dev)
        stream_ptr = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(L.pg_mini_gather_entries(loc.desc(), loc._half_ws.data_ptr(), loc._half_ws.numel(), fill.data_ptr(), elem.data_ptr(),
                                                send.data_ptr(), send.numel(), loc.status.data_ptr(), stream_ptr))
        recv = torch.empty(world * cap1, dtype=torch.int64, device=dev)
        _all_to_all_flat(recv, send, group)
        mine = (cuts[me], cuts[me + 1])
        seg_me = (seg[:, mine[0]:mine[1] + 1] - at[:, me:me + 1]).contiguous()        # [part, owned + 1]
        bins_out = torch.empty(world * cap1, dtype=torch.int16, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.pg_mini_merge_bins(recv.data_ptr(), cap1, seg_me.data_ptr(), world, uni.desc(), mine[0], mine[1], self.window, self.vsize,
                                            bins_out.data_ptr(), uni.status.data_ptr(), stream_ptr))
        uni._empty = False
        bins = torch.empty(world * cap1, dtype=torch.int16, device=dev)
        _all_to_all_flat(bins.view(torch.uint8), bins_out.view(torch.uint8), group)     # (as bytes: neither RCCL nor gloo moves int16)
        self._bins, self._elem = bins, elem
        self.bytes_sent = self.bytes_received = (8 + 2) * cap1 * (world - 1)         # (padded to the longest part: what the collectives move)

    def lookup_half(self) -> None:
        self.local.lookup_half(self._bins, self._elem)
        self._bins = self._elem = None

    def status_bits(self) -> int:
        """the PG_STATUS_* bits any rank's kernels raised (the rank's own count / lookup half, or an owner's merge): the SAME word
        everywhere -- one all-reduce (MAX) over the bits, taken apart"""
        st = (self.local.status[:1] | self.union.status[:1]).to(torch.int64)
        bits = ((st >> torch.arange(4, device=st.device)) & 1).to(torch.int32)
        bits = _staged(bits, self.group)
        dist.all_reduce(bits, op=dist.ReduceOp.MAX, group=self.group)
        return int(sum(int(b) << i for i, b in enumerate(bits.tolist())))

    def any_full(self, bits: int | None = None) -> bool:
        """did a bucket run full on ANY rank (the rank's own LDS table, or an owner's merged one)?  The same answer everywhere.
        Anything else the kernels reported (a plan that does not describe the stream: nothing was counted; a store outside a
        buffer in the checked build) is raised -- on every rank -- instead of being taken for a result."""
        from . import _lib
        bits = self.status_bits() if bits is None else bits
        if bits & _lib.STATUS_BOUNDS:
            raise RuntimeError("a kernel of the checked build was about to store outside its buffer (PG_STATUS_BOUNDS) on some rank: the results are incomplete")
        if bits & _lib.STATUS_PLAN_MISMATCH:
            self.local._mini_plan = None
            raise RuntimeError("the partition plan did not describe the stream on some rank (PG_STATUS_PLAN_MISMATCH): nothing was counted")
        if bits & _lib.STATUS_OVERFLOW_LIST:
            self._cap1 = None
            raise RuntimeError("a part of the exchange was longer than the size kept from an earlier batch (PG_STATUS_OVERFLOW_LIST): nothing was "
                               "exchanged -- count again (MiniSharded.count(check=True) does)")
        return bool(bits & _lib.STATUS_TABLE_FULL)

    def check_status(self, bits: int | None = None) -> None:
        from . import _lib
        if self.any_full(bits):
            raise _lib.PangaeaError(_lib.PG_ETABLEFULL, "a bucket of the local or of the merged table is full")

    def owned_items(self):
        """(codes, counts) of the buckets this rank owns, sorted by code -- host copies, for tests"""
        world, me = dist.get_world_size(self.group), dist.get_rank(self.group)
        nb = self.union.n_buckets
        b0, b1 = nb * me // world, nb * (me + 1) // world
        from . import _lib
        sl = self.union.data.view(nb, -1)[b0:b1].reshape(-1)
        x = sl[sl != 0].cpu().numpy().view(np.uint64)
        codes, counts = x >> np.uint64(_lib.HASH_COUNT_BITS), x & np.uint64((1 << _lib.HASH_COUNT_BITS) - 1)
        order = np.argsort(codes)
        return codes[order], counts[order]


def features_sharded_mini(stream: ReadStream, plan, k: int, k_tnf: int | None, window: int, vsize: int, group=None, max_tries: int = 4):
    """(tnf, abd, MiniSharded) of this rank's rows with the k-mers of ALL ranks' reads counted: sketches -> geometry -> count,
    exchange, lookups; a full bucket anywhere enlarges the geometry on every rank and counts again"""
    from . import kmer
    regs = kmer.distinct_sketch(stream, k)
    local = kmer.sketch_estimate(regs)
    union = _staged(regs.clone(), group)
    dist.all_reduce(union, op=dist.ReduceOp.MAX, group=group)
    total = max(1 << 13, int(1.05 * kmer.sketch_estimate(union)))
    # (every rank must pick the same geometry: the largest local estimate decides the local bucket size)
    loc = torch.tensor([int(1.1 * local)], dtype=torch.int64)
    loc = loc.to(stream.device) if dist.get_backend(group) == "nccl" else loc
    dist.all_reduce(loc, op=dist.ReduceOp.MAX, group=group)
    from . import _lib
    log2_u, lb_u, lb_l = MiniSharded.geometry(total, int(loc.item()))

    def key_partitioned():
        # a union that the super-k-mer geometry cannot hold (more than 2^16 buckets of 2^14 slots): the key-partitioned exchange,
        # on every rank -- the estimates behind this decision were all-reduced, so it is the same everywhere
        table = count_kmers_sharded(stream, k, rows=plan, group=group)
        tnf, abd = kmer.features(stream, plan, k_tnf=k_tnf, table=table, window=window, vsize=vsize)
        return tnf, abd, None

    for attempt in range(max_tries):
        if log2_u - lb_u > _lib.MINI_MAX_LOG2_BUCKETS:
            return key_partitioned()
        ms = MiniSharded(k, stream.device, log2_u, lb_l, window, vsize, group, union_log2_bucket=lb_u)
        ms.count(stream, plan, check=False)
        if not ms.any_full():
            tnf, abd = kmer.features(stream, plan, k_tnf=k_tnf, table=ms.local, window=window, vsize=vsize)
            return tnf, abd, ms
        del ms
        lb_l = min(lb_l + 1, lb_u)
        log2_u += 1
        lb_u = min(_lib.BUCKET_MAX_LOG2_SLOTS, lb_u + 1)
    raise _lib.PangaeaError(_lib.PG_ETABLEFULL, "the sharded super-k-mer tables stayed full")
