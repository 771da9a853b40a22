"""Device tables of global canonical k-mer multiplicities and the per-run feature rows.

Mirrors, in one process and on the GPU, what ``src/feature.py`` obtains from three subprocesses:
``jellyfish count/dump`` (feature.py:94,103) -> :class:`KmerTable`; ``count_tnf`` (feature.py:133) and
``count_kmer`` (feature.py:109) -> :func:`features`.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from .reads import ReadStream, Rows

DEFAULT_SEG_CHARS = 65536              # row segments of the lookup kernels (K1: every segment ends in 136 global adds; 16384: 0.98 ms, 65536: 0.83 ms at 10 M pairs)


def _stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on a GPU: the k-mer kernels have no CPU path "
                           f"(got a tensor on {t.device})")


class KmerTable:
    """exact multiplicity of every canonical k-mer over every read of the input.

    ``dense``: int32 tensor of 4^k counters (k <= 16).  ``hash``: int64 tensor of 2^log2_slots slots,
    slot = (key42(code) << 22) | count, 0 = empty (k <= 21), optionally split into buckets of 2^log2_bucket slots
    (probing wraps inside a bucket); bucketed tables are built by the partition + LDS-count pipeline
    (``pg_kmer_count_bucketed``), unbucketed ones by one global atomic per occurrence (``pg_kmer_count``).
    ``wide`` (22 <= k <= 31, the rest of the reference's range): 2^log2_slots int64 keys (code + 1) followed by as
    many int32 counts in the same tensor; direct kernels only.
    """

    # bytes of scratch one bucketed launch may use (longer streams are counted in pieces): two 8-byte record buffers now, the
    # row shuffle adds half as much again later -- 132 GiB + 66 GiB + table + stream stay inside 288 GB of HBM and cover the
    # 25 M-pair share of BASELINE config 3 in one piece
    WORKSPACE_BUDGET = 132 << 30

    def __init__(self, k: int, kind: str, data: torch.Tensor, log2_slots: int = 0, log2_bucket: int = 0):
        self.k, self.kind, self.data, self.log2_slots, self.log2_bucket = int(k), kind, data, int(log2_slots), int(log2_bucket)
        self.status = torch.zeros(2, dtype=torch.int32, device=data.device)
        code = {"dense": _lib.TABLE_DENSE, "hash": _lib.TABLE_HASH, "wide": _lib.TABLE_WIDE, "mini": _lib.TABLE_MINI,
                "miniw": _lib.TABLE_MINI_WIDE}[kind]
        self._desc = _lib.pg_table(code, self.k, self.log2_slots, self.log2_bucket, data.data_ptr())
        self._empty = True               # nothing counted since allocation / reset()
        self._workspace = None
        self._shuffle_ws = None
        self._records = None             # (plan, n_words) while the workspace holds the row-tagged records of ONE count
        self._deferred = None            # (fill, n_words) after a deferred count: entries wait in the workspace, slots unwritten
        self._emitted = None             # (window, vsize) while the shuffle workspace holds the words of a fused count + lookup
        self._mini_plan = None           # (key, plan workspace, n_records) of the last pg_mini_plan: reused while the key matches
        self._mini_next = None           # (key, plan workspace, event, rows) of a plan computed ahead on a side stream
        self._mini_spare = None          # the plan workspace that is neither in use nor being filled
        self._mini_rec_ws = None
        self._mini_sized_for = None      # (n_words, geometry) the record / slot workspaces were sized for (with slack)
        self._mini_optimistic = None     # the arguments of a count that ran on them without reading its plan's counts
        self._mini_pieces = 1            # word ranges the last count of a mini table was done in (``_count_mini_pieces``)
        self._half = None                # (fill, n_words, rows, window, vsize) between count_half and lookup_half (N > 1 ranks)
        self._half_ws = None
        self._merge_ws = None            # the provisional words of the merged lookups (fixed slots per record)

    # ------------------------------------------------------------------ construction

    @staticmethod
    def default_kind(k: int) -> str:
        if k < 1 or k > _lib.WIDE_MAX_K:
            raise ValueError(f"k-mer size {k} unsupported (1..{_lib.WIDE_MAX_K}, as the reference)")
        if k > _lib.HASH_MAX_K:
            return "wide"
        # measured at 10 M pairs: k=15 dense 191 ms vs hash (partition + LDS) 70 ms; k=11 135 vs 69 ms.  Dense tables only
        # where 4^k counters stay cache resident.
        return "dense" if k <= 8 else "hash"

    @staticmethod
    def default_log2_bucket(log2_slots: int) -> int:
        """bucket size for a table of 2^log2_slots slots, 0 when the table cannot be bucketed: buckets hold at most
        2^14 slots (LDS) and there are at most 2^15 of them (two scatter passes)"""
        if log2_slots < 14:
            return 0                                       # small tables: the direct kernel is as good
        lb = min(_lib.BUCKET_MAX_LOG2_SLOTS, max(10, log2_slots - 12))
        return lb if 1 <= log2_slots - lb <= _lib.BUCKET_MAX_LOG2_BUCKETS else 0

    @classmethod
    def alloc(cls, k: int, device, kind: str | None = None, distinct_hint: int | None = None,
              load: float = 0.5, log2_bucket: int | None = None) -> "KmerTable":
        kind = kind or cls.default_kind(k)
        device = torch.device(device)
        if kind == "dense":
            if k > _lib.DENSE_MAX_K:
                raise ValueError(f"dense tables need k <= {_lib.DENSE_MAX_K}")
            return cls(k, "dense", torch.zeros(4 ** k, dtype=torch.int32, device=device))
        if kind in ("mini", "miniw"):
            want = max(1024, int((distinct_hint or 1 << 20) / load))
            return cls.mini_with_slots(k, device, max(10, math.ceil(math.log2(want))), log2_bucket)
        if kind not in ("hash", "wide"):
            raise ValueError(f"unknown table kind {kind!r}")
        if kind == "hash" and k > _lib.HASH_MAX_K:
            raise ValueError(f"hash tables need k <= {_lib.HASH_MAX_K}")
        want = max(1024, int((distinct_hint or 1 << 20) / load))
        log2 = max(10, math.ceil(math.log2(want)))
        if kind == "wide":
            return cls.wide_with_slots(k, device, log2)
        return cls.with_slots(k, device, log2, log2_bucket)

    @staticmethod
    def mini_max_log2_bucket(k: int) -> int:
        """slots of one LDS-resident bucket: 8-byte packed slots up to k = 21, 8-byte keys + 4-byte counts beyond"""
        return _lib.BUCKET_MAX_LOG2_SLOTS if k <= _lib.HASH_MAX_K else _lib.MINI_WIDE_MAX_LOG2_BUCKET_SLOTS

    @staticmethod
    def mini_applies(k: int, log2_slots: int, log2_bucket: int | None = None) -> bool:
        """can a MINI table (minimizer buckets, built from super-k-mers) hold 2^log2_slots slots for this k?"""
        if not _lib.MINI_MIN_K <= k <= _lib.WIDE_MAX_K:
            return False
        top = KmerTable.mini_max_log2_bucket(k)
        lb = min(top, log2_slots) if log2_bucket is None else log2_bucket
        return 4 <= lb <= top and 0 <= log2_slots - lb <= _lib.MINI_MAX_LOG2_BUCKETS

    @classmethod
    def mini_with_slots(cls, k: int, device, log2_slots: int, log2_bucket: int | None = None) -> "KmerTable":
        """``mini`` (k <= 21: packed 8-byte slots) or ``miniw`` (22 <= k <= 31: keys + counts planes, as ``wide``)"""
        top = cls.mini_max_log2_bucket(k)
        lb = min(top, log2_slots) if log2_bucket is None else log2_bucket
        want = os.environ.get("PG_MINI_LOG2_BUCKET")            # tuning / comparison: bucket size of tables whose caller named none
        if log2_bucket is None and want and cls.mini_applies(k, log2_slots, min(int(want), lb)):
            lb = min(int(want), lb)
        if not cls.mini_applies(k, log2_slots, lb):
            raise ValueError(f"mini tables need {_lib.MINI_MIN_K} <= k <= {_lib.WIDE_MAX_K} and at most 2^{_lib.MINI_MAX_LOG2_BUCKETS} buckets "
                             f"of at most 2^{top} slots (k {k}, 2^{log2_slots} slots, buckets of 2^{lb})")
        n = 1 << log2_slots
        if k > _lib.HASH_MAX_K:
            return cls(k, "miniw", torch.zeros(n + n // 2, dtype=torch.int64, device=device), log2_slots, lb)
        return cls(k, "mini", torch.zeros(n, dtype=torch.int64, device=device), log2_slots, lb)

    @classmethod
    def wide_with_slots(cls, k: int, device, log2_slots: int) -> "KmerTable":
        n = 1 << log2_slots
        return cls(k, "wide", torch.zeros(n + n // 2, dtype=torch.int64, device=device), log2_slots, 0)

    def _wide_parts(self):
        n = 1 << self.log2_slots
        return self.data[:n], self.data[n:].view(torch.int32)

    @classmethod
    def with_slots(cls, k: int, device, log2_slots: int, log2_bucket: int | None = None) -> "KmerTable":
        lb = cls.default_log2_bucket(log2_slots) if log2_bucket is None else log2_bucket
        return cls(k, "hash", torch.zeros(1 << log2_slots, dtype=torch.int64, device=device), log2_slots, lb)

    @classmethod
    def from_items(cls, k: int, codes, counts, device, kind: str | None = None) -> "KmerTable":
        """table holding exactly the given (canonical code, count) entries -- e.g. a parsed jellyfish dump
        (count_kmer.cpp:139-170 assigns, it does not add: later duplicates must already be resolved)"""
        codes = torch.as_tensor(np.asarray(codes).astype(np.int64))
        counts = torch.as_tensor(np.asarray(counts).astype(np.int64))
        table = cls.alloc(k, device, kind, distinct_hint=max(1024, codes.numel()))
        if table.kind in ("wide", "miniw"):
            c = codes.to(table.device).contiguous()
            n = counts.to(table.device, torch.int32).contiguous()
            table._empty = False
            with torch.cuda.device(table.device):
                _lib.check(_lib.load().pg_kmer_merge_wide(c.data_ptr(), n.data_ptr(), c.numel(), table.desc(), table.status.data_ptr(),
                                                          _stream_ptr(table.device)))
            table.check_status()
        elif table.kind == "dense":
            table.data[codes.to(table.device)] = counts.to(table.device, torch.int32)
        elif table.kind == "mini":                     # the slots hold the codes themselves
            sat = torch.clamp(counts, max=_lib.HASH_COUNT_SAT)
            table.merge(((codes << _lib.HASH_COUNT_BITS) | sat)[sat > 0])
        else:
            sat = torch.clamp(counts, max=_lib.HASH_COUNT_SAT)
            keys = torch.from_numpy(key42(codes.numpy().view(np.uint64)).view(np.int64))
            table.merge(((keys << _lib.HASH_COUNT_BITS) | sat)[sat > 0])
        return table

    @property
    def device(self) -> torch.device:
        return self.data.device

    @property
    def nbytes(self) -> int:
        return self.data.numel() * self.data.element_size()

    def desc(self):
        return C.byref(self._desc)

    # ------------------------------------------------------------------ counting

    def reset(self) -> "KmerTable":
        """forget every count.  Bucketed tables are not even cleared: the next count overwrites every slice."""
        if not (self.kind in ("hash", "mini", "miniw") and self.log2_bucket):
            self.data.zero_()
        self.status.zero_()
        self._empty = True
        self._records = None
        self._deferred = None
        self._emitted = None
        return self

    def _workspace_for(self, n_words: int) -> torch.Tensor:
        need = _lib.check(_lib.load().pg_kmer_count_workspace_bytes(n_words, self.desc()))
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._workspace

    def can_defer(self, n_words: int) -> bool:
        """may ``count(..., deferred_group=g)`` be used for a fresh count of ``n_words`` words?"""
        step = int(self.WORKSPACE_BUDGET // (2 * 8 * 32))
        return self._bucketed() and self._empty and self.log2_slots - self.log2_bucket > 8 and n_words <= step

    def _shuffle_workspace_for(self, n_words: int, n_rows: int, vsize: int) -> torch.Tensor:
        need = _lib.check(_lib.load().pg_abundance_workspace_bytes(n_words, n_rows, vsize, self.desc()))
        if self._shuffle_ws is None or self._shuffle_ws.numel() < need:
            self._shuffle_ws = None
            self._shuffle_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._shuffle_ws

    def count(self, stream: ReadStream, word_begin: int = 0, word_end: int | None = None, check: bool = True,
              rows: "Plan | None" = None, deferred_group: int | None = None, emit: tuple | None = None,
              lowercase_is_base: bool = False) -> "KmerTable":
        """add the k-mers ending in words [word_begin, word_end) of the stream (asynchronous unless ``check``).
        With ``rows`` (a Plan of this stream's rows) a bucketed table also keeps the row-tagged partition records, which
        lets ``features`` build the abundance rows by shuffle instead of by table lookups.

        ``deferred_group`` = g (multi-GPU, ``can_defer``): this table has the geometry of the union over all ranks and
        is NOT written; 2^g adjacent buckets are counted together in LDS and only their occupied entries and the
        per-bucket fills are kept, for ``dist.exchange_table`` to gather and to rebuild the table from.  Until then the
        table holds no counts (``pending``).

        ``emit`` = (window, vector_size) (one GPU, a fresh table, ``rows`` given, at least 2^11 buckets): the lookup pass of
        the abundance rows is fused into the counting kernel -- a bucket's records are looked up while its counts are
        still in LDS -- and ``features`` with the same window and vector size starts from the emitted (row, bin) words.
        Silently ignored where it does not apply.

        ``lowercase_is_base``: count lower-case a c g t as bases, as ``jellyfish count`` does (src/feature.py:94) -- the
        reference's own row counters reset on them (count_kmer.cpp:73-78), so k-mers that are only valid under this rule
        enter the table but belong to no row.  Only matters for soft-masked input (``stream.valid_lower`` is not None).
        Paired input with bases below the quality threshold (``stream.valid_lowq``, jellyfish's --min-qual-char=? of
        feature.py:76-83) is always counted without them, and without ``rows`` / ``emit`` (see ``ReadStream.table_valid``)."""
        _require_gpu(stream.codes, "the read stream")
        if stream.device != self.device:
            raise ValueError("stream and table are on different devices")
        word_end = stream.n_words if word_end is None else word_end
        L = _lib.load()
        # the table is counted with jellyfish's view of the reads (lower-case bases count when asked for, bases below the
        # quality threshold of paired input never do); rows keep the reference's own rule, the strict plane
        table_plane = stream.table_valid(lowercase_is_base)
        lenient = table_plane is not stream.valid
        valid_ptr = table_plane.data_ptr()
        if not stream.rows_inside_table:
            # quality-masked bases: a row's k-mer may be missing from the table, which only the lookup form of ``features``
            # expresses (count_kmer.cpp:87) -- no row-tagged records, no fused lookups
            rows = emit = None

        def rows_arg(plan):
            # the rows' own validity rule stays the strict one: with a lenient counting plane the strict plane rides along
            if plan is None:
                return None
            if not lenient:
                return C.byref(plan.rows_desc)
            desc = _lib.pg_rows(plan.row_start.data_ptr(), plan.row_end.data_ptr(), plan.n_rows, stream.valid.data_ptr())
            self._rows_desc_keepalive = desc
            return C.byref(desc)

        if self.kind in ("mini", "miniw"):
            if deferred_group is not None:
                raise ValueError("mini tables have no deferred form")
            return self._count_mini(stream, word_begin, word_end, table_plane, rows, rows_arg, emit, lenient, check)
        if deferred_group is not None:
            if not self.can_defer(word_end - word_begin):
                raise ValueError("deferred counting needs a fresh bucketed table with more than 256 buckets and a single pass")
            g = max(0, min(int(deferred_group), _lib.DEFERRED_MAX_GROUP_LOG2, self.log2_slots - self.log2_bucket - 8))
            keep = rows if (rows is not None and rows.shuffle_ok) else None
            ws = self._workspace_for(word_end - word_begin)
            fill = torch.empty(self.n_buckets, dtype=torch.int64, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(L.pg_kmer_count_deferred(stream.codes.data_ptr(), valid_ptr, word_begin, word_end, self.desc(), g,
                                                    rows_arg(keep), ws.data_ptr(), ws.numel(),
                                                    fill.data_ptr(), self.status.data_ptr(), _stream_ptr(self.device)))
            self._empty = False
            self._deferred = (fill, word_end - word_begin)
            self._records = (keep, word_end - word_begin) if keep is not None else None
            if check:
                self.check_status()
            return self
        self._deferred = None
        self._emitted = None
        with torch.cuda.device(self.device):
            if self.kind == "hash" and self.log2_bucket:
                # pieces bounded by the scratch budget: two record buffers of 8 B per character
                step = max(_lib.WORD_ALIGN, int(self.WORKSPACE_BUDGET // (2 * 8 * 32)) // _lib.WORD_ALIGN * _lib.WORD_ALIGN)
                single = self._empty and word_end - word_begin <= step
                keep = rows if (single and rows is not None and rows.shuffle_ok) else None
                self._records = None
                if (emit is not None and keep is not None and self._bucketed() and self.tag_bits <= 31 and word_end > word_begin
                        and 1 <= emit[1] <= _lib.SHUFFLE_MAX_VSIZE and emit[0] >= 1 and emit[0] * emit[1] <= _lib.HASH_COUNT_SAT):
                    window, vsize = int(emit[0]), int(emit[1])
                    n_words = word_end - word_begin
                    ws = self._workspace_for(n_words)
                    sws = self._shuffle_workspace_for(n_words, keep.n_rows, vsize)
                    _lib.check(L.pg_kmer_count_bucketed_emit(stream.codes.data_ptr(), valid_ptr, word_begin, word_end, self.desc(),
                                                             rows_arg(keep), ws.data_ptr(), ws.numel(), window, vsize,
                                                             sws.data_ptr(), sws.numel(), self.status.data_ptr(), _stream_ptr(self.device)))
                    self._empty = False
                    self._records = (keep, n_words)
                    self._emitted = (window, vsize)
                    if check:
                        self.check_status()
                    return self
                for w0 in range(word_begin, word_end, step):
                    w1 = min(word_end, w0 + step)
                    ws = self._workspace_for(w1 - w0)
                    _lib.check(L.pg_kmer_count_bucketed(stream.codes.data_ptr(), valid_ptr, w0, w1, self.desc(),
                                                        0 if self._empty else 1, rows_arg(keep),
                                                        ws.data_ptr(), ws.numel(), self.status.data_ptr(), _stream_ptr(self.device)))
                    self._empty = False
                if keep is not None:
                    self._records = (keep, word_end - word_begin)
            else:
                _lib.check(L.pg_kmer_count(stream.codes.data_ptr(), valid_ptr, word_begin, word_end,
                                           self.desc(), self.status.data_ptr(), _stream_ptr(self.device)))
                self._empty = False
        if check:
            self.check_status()
        return self

    @staticmethod
    def _plan_key(codes: torch.Tensor, plane: torch.Tensor, word_begin: int, word_end: int, keep, lenient: bool, log2_slots: int, log2_bucket: int):
        """what a cached partition plan is valid for.  Addresses alone would not do: after the stream is freed the caching
        allocator hands the same address to the next stream of that size, and tensors can be rewritten in place -- so the key
        carries the tensors' version counters, and the cache entry holds the tensors (and the rows) themselves, which keeps
        their storage from being recycled while the plan is kept.  (pg_mini_count checks the plan's record count against the
        record workspace as well: PG_STATUS_PLAN_MISMATCH.)"""
        return (codes.data_ptr(), codes._version, plane.data_ptr(), plane._version, word_begin, word_end, id(keep), bool(lenient),
                log2_slots, log2_bucket)

    def count_half(self, stream: ReadStream, rows: "Plan", emit: tuple, check: bool = True) -> "KmerTable":
        """N > 1 ranks (``dist.MiniSharded``): the COUNT half of the super-k-mer pipeline on this rank's reads.  This table object
        only carries the rank's LOCAL geometry (the union's bucket count, slots for the rank's own k-mers): its slots are never
        written.  Left behind for ``dist``: the buckets' entries and occupancy (``_half``), the provisional words of the rows."""
        if self.kind != "mini":
            raise ValueError("count_half() is for packed mini tables (13 <= k <= 21)")
        _require_gpu(stream.codes, "the read stream")
        plane = stream.table_valid(False)
        if plane is not stream.valid or not stream.rows_inside_table:
            raise ValueError("count_half() takes plain streams (no soft-masked / quality-masked planes)")
        return self._count_mini(stream, 0, stream.n_words, plane, rows, lambda plan: C.byref(plan.rows_desc), emit, False, check, half=True)

    def _count_mini(self, stream, word_begin, word_end, table_plane, rows, rows_arg, emit, lenient, check, half=False):
        """the super-k-mer pipeline (pg_mini_plan + pg_mini_count): a fresh table, one piece.  The partition plan depends on
        the stream, the rows and the geometry only and is kept: counting the same range again skips pg_mini_plan."""
        valid_ptr = table_plane.data_ptr()
        if not self._empty:
            raise ValueError("mini tables are built by ONE count of a fresh (or reset) table")
        L = _lib.load()
        n_words = word_end - word_begin
        keep = rows if (rows is not None and rows.shuffle_ok and rows.n_rows <= _lib.MINI_MAX_ROWS) else None
        if rows is not None and keep is None:
            raise ValueError("mini tables need sorted, disjoint, non-empty rows (at most 2^20 - 2 of them)")
        fuse = (emit is not None and keep is not None and n_words > 0 and 1 <= emit[1] <= _lib.SHUFFLE_MAX_VSIZE and emit[0] >= 1
                and (self.kind == "miniw" or emit[0] * emit[1] <= _lib.HASH_COUNT_SAT))
        # a stream whose scratch would not fit in one piece (about 3.4 KB per 150 bp read pair; PANGAEA_MINI_PIECE_WORDS forces a
        # piece size) is counted word range by word range into the same table, its lookups done when the table is final
        piece_words = self._piece_words(n_words, fuse and not half)
        if piece_words is not None:
            return self._count_mini_pieces(stream, word_begin, word_end, table_plane, keep, rows_arg, emit, piece_words, check)
        key = self._plan_key(stream.codes, table_plane, word_begin, word_end, keep, lenient, self.log2_slots, self.log2_bucket)
        held = (stream.codes, table_plane)               # (kept with the plan: see _plan_key)
        with torch.cuda.device(self.device):
            if (self._mini_plan is None or self._mini_plan[0] != key) and self._mini_next is not None and self._mini_next[0] == key:
                # a plan computed ahead (``prefetch_plan``): the count waits for it ON THE DEVICE.  Its record counts size the
                # record and slot workspaces -- but where workspaces of a batch of the same size exist already (a stream of batches),
                # they are used as they are and the host does not wait for the plan at all: the kernels themselves refuse a plan that
                # names more records than the buffers hold (PG_STATUS_PLAN_MISMATCH, nothing is written), and ``check_status`` /
                # the check below then count again with workspaces of the right size.  The counts stay on the device until
                # somebody asks (``plan_counts``).
                _, ws, event, _ = self._mini_next[:4]
                torch.cuda.current_stream(self.device).wait_event(event)
                if self._mini_plan is not None:
                    self._mini_spare = self._mini_plan[1]
                sized_for = getattr(self, "_mini_sized_for", None)
                if (sized_for == (n_words, self.log2_slots, self.log2_bucket) and self._mini_rec_ws is not None
                        and os.environ.get("PG_PLAN_HOST_SYNC", "0") in ("", "0")):
                    self._mini_plan = (key, ws, None, keep, held, None)
                else:
                    event.synchronize()
                    head = ws[:24].view(torch.int64).cpu()                 # (records, -, records of more than four k-mers)
                    self._mini_plan = (key, ws, int(head[0]), keep, held, int(head[2]))
                self._mini_next = None
            if self._mini_plan is None or self._mini_plan[0] != key:
                need = _lib.check(L.pg_mini_plan_bytes(n_words, self.desc()))
                ws = self._mini_spare if self._mini_spare is not None and self._mini_spare.numel() == need else None
                self._mini_spare = None
                if ws is None:
                    ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                _lib.check(L.pg_mini_plan(stream.codes.data_ptr(), valid_ptr, word_begin, word_end, self.desc(), rows_arg(keep),
                                          ws.data_ptr(), ws.numel(), _stream_ptr(self.device)))
                head = ws[:24].view(torch.int64).cpu()                     # (host sync; once per plan: records, -, long records)
                n_records = int(head[0])
                if self._mini_plan is not None:
                    self._mini_spare = self._mini_plan[1]
                self._mini_plan = (key, ws, n_records, keep, held, int(head[2]))
            _, plan_ws, n_records = self._mini_plan[:3]
            optimistic = n_records is None          # (workspaces of the previous batch of this size, the plan's counts unread)
            # (3 % of slack on the counts: the next batch of the same size then finds room without asking)
            slack = (lambda n: n + n // 32 + 4096)
            if not optimistic:
                need = _lib.check(L.pg_mini_records_bytes(slack(n_records), self.desc()))
                if self._mini_rec_ws is None or self._mini_rec_ws.numel() < need or self._mini_rec_ws.numel() > 2 * need:
                    self._mini_rec_ws = None
                    self._mini_rec_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                self._mini_sized_for = (n_words, self.log2_slots, self.log2_bucket)
            window, vsize, sws_ptr, sws_n = 0, 0, None, 0
            if fuse:
                window, vsize = int(emit[0]), int(emit[1])
                # (with the merged lookups the provisional data live in their own buffer below: the row shuffle's layout is smaller)
                merging = os.environ.get("PG_MINI_MERGE", "1") not in ("", "0")
                need = _lib.check(L.pg_mini_shuffle_bytes_merged(n_words, keep.n_rows, vsize, self.desc()) if merging
                                  else L.pg_mini_shuffle_bytes(n_words, keep.n_rows, vsize))
                if self._shuffle_ws is None or self._shuffle_ws.numel() < need:
                    self._shuffle_ws = None
                    self._shuffle_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                sws_ptr, sws_n = self._shuffle_ws.data_ptr(), self._shuffle_ws.numel()
            mws_ptr, mws_n = None, 0
            if fuse and os.environ.get("PG_MINI_MERGE", "1") not in ("", "0"):
                # the merged form of the lookups (PG_MINI_MERGE=0: word-wise): its provisional words lie in fixed slots per record,
                # sized from the plan's record counts (records, and records of more than four k-mers: 1st and 3rd word of the plan
                # workspace); the library falls back to the word-wise form where the merged one does not apply
                if not optimistic or getattr(self, "_merge_ws", None) is None:
                    if optimistic:                                         # (no slot buffer yet: the counts are needed after all)
                        n_records, n_long = self.plan_counts()
                        optimistic = False
                    else:
                        n_long = self._mini_plan[5]
                    need = _lib.check(L.pg_mini_merge_words(n_words, slack(n_records), min(slack(n_long), slack(n_records)), self.desc()))
                    if getattr(self, "_merge_ws", None) is None or self._merge_ws.numel() < need:
                        self._merge_ws = None
                        self._merge_ws = torch.empty(need, dtype=torch.int32, device=self.device)
                mws_ptr, mws_n = self._merge_ws.data_ptr(), self._merge_ws.numel()
            if half:
                if not fuse:
                    raise ValueError("count_half() needs rows and abundance parameters")
                need = _lib.check(L.pg_mini_half_bytes(self.desc()))
                if getattr(self, "_half_ws", None) is None or self._half_ws.numel() != need:
                    self._half_ws = None
                    self._half_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                # (zeroed: a count half that refuses its plan -- PG_STATUS_PLAN_MISMATCH -- writes nothing, and the exchange sizes its
                # buffers from these numbers before anybody has looked at the status word)
                fill = torch.zeros(self.n_buckets, dtype=torch.int64, device=self.device)
                _lib.check(L.pg_mini_count_half(stream.codes.data_ptr(), valid_ptr, word_begin, word_end, self.desc(), rows_arg(keep),
                                                plan_ws.data_ptr(), plan_ws.numel(), self._mini_rec_ws.data_ptr(), self._mini_rec_ws.numel(),
                                                window, vsize, sws_ptr, sws_n, mws_ptr, mws_n, self._half_ws.data_ptr(), self._half_ws.numel(),
                                                fill.data_ptr(), self.status.data_ptr(), _stream_ptr(self.device)))
                self._half = (fill, n_words, keep, window, vsize)
            else:
                _lib.check(L.pg_mini_count(stream.codes.data_ptr(), valid_ptr, word_begin, word_end, self.desc(), rows_arg(keep),
                                           plan_ws.data_ptr(), plan_ws.numel(), self._mini_rec_ws.data_ptr(), self._mini_rec_ws.numel(),
                                           window, vsize, sws_ptr, sws_n, mws_ptr, mws_n, self.status.data_ptr(), _stream_ptr(self.device)))
        self._empty = False
        self._mini_pieces = 1
        self._records = (keep, n_words) if fuse and not half else None
        self._emitted = (window, vsize) if fuse and not half else None
        self._mini_optimistic = (stream, word_begin, word_end, rows, emit, half) if optimistic else None
        if check:
            self.check_status()
        return self

    # bytes of scratch per word of the stream (32 characters): record workspace (two planes of 12-byte records, ~6.7 records per
    # word at k = 21), 2-byte slots of the merged lookups, the row shuffle's word regions (4 bytes per character)
    _PIECE_BYTES_PER_WORD = (24 * 7, 12 * 7, 4 * 32)

    def _piece_words(self, n_words: int, applicable: bool):
        """None (one piece, the usual path) or the number of words per piece"""
        forced = os.environ.get("PANGAEA_MINI_PIECE_WORDS")
        if not applicable or self.kind != "mini" or self.n_buckets <= 256 or os.environ.get("PG_MINI_MERGE", "1") in ("", "0"):
            return None                                          # (the pieces' kernels: packed slots, both scatter passes, the merged lookups)
        if forced:
            w = max(_lib.WORD_ALIGN, int(forced) // _lib.WORD_ALIGN * _lib.WORD_ALIGN)
            return w if w < n_words else None
        rec, slots, words = self._PIECE_BYTES_PER_WORD
        free, _ = torch.cuda.mem_get_info(self.device)
        free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)      # (cached blocks count as free)
        budget = 0.85 * free
        if (rec + slots + words) * n_words <= budget:
            return None
        room = budget - (slots + words + 4 * 7) * n_words          # what stays per word whatever the piece size (slots, kept meta, words)
        if room <= rec * 4 * _lib.WORD_ALIGN:
            return None                                          # (not even in pieces: the one-piece path reports the allocation that fails)
        return max(_lib.WORD_ALIGN, int(room / rec) // _lib.WORD_ALIGN * _lib.WORD_ALIGN)

    def _count_mini_pieces(self, stream, word_begin, word_end, table_plane, keep, rows_arg, emit, piece_words, check):
        """``_count_mini`` for a stream counted in word ranges of ``piece_words`` (include/pangaea_feat.h: pg_mini_count_piece):
        every piece -> its plan, both scatter passes, the count INTO the table's buckets (slots keep their places), its 2-byte
        provisional slots and its records' meta words kept; then the lookups of every piece in the final table, into the row
        shuffle's regions of the whole stream.  Afterwards the table is what one count would have made it (``features`` reads
        the rows from the shuffled words as usual)."""
        L = _lib.load()
        valid_ptr = table_plane.data_ptr()
        window, vsize = int(emit[0]), int(emit[1])
        n_words = word_end - word_begin
        ranges = [(w0, min(word_end, w0 + piece_words)) for w0 in range(word_begin, word_end, piece_words)]
        slack = (lambda n: n + n // 32 + 4096)
        kept = []
        sp = _stream_ptr(self.device)
        with torch.cuda.device(self.device):
            rec_ws = None
            for idx, (w0, w1) in enumerate(ranges):
                need = _lib.check(L.pg_mini_plan_bytes(w1 - w0, self.desc()))
                plan_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                _lib.check(L.pg_mini_plan(stream.codes.data_ptr(), valid_ptr, w0, w1, self.desc(), rows_arg(keep), plan_ws.data_ptr(), plan_ws.numel(), sp))
                head = plan_ws[:24].view(torch.int64).cpu()                 # (host wait, once per piece: records, -, long records)
                n_records, n_long = int(head[0]), int(head[2])
                need = _lib.check(L.pg_mini_records_bytes(slack(n_records), self.desc()))
                if rec_ws is None or rec_ws.numel() < need:
                    rec_ws = None
                    rec_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                need = _lib.check(L.pg_mini_merge_words(w1 - w0, n_records, n_long, self.desc()))
                merge_ws = torch.empty(need, dtype=torch.int32, device=self.device)
                _lib.check(L.pg_mini_count_piece(stream.codes.data_ptr(), valid_ptr, w0, w1, self.desc(), rows_arg(keep), plan_ws.data_ptr(), plan_ws.numel(),
                                                 rec_ws.data_ptr(), rec_ws.numel(), window, vsize, merge_ws.data_ptr(), merge_ws.numel(),
                                                 1 if idx == 0 else 0, self.status.data_ptr(), sp))
                # the bucket-ordered records' meta words (lengths, rows): the second meta plane of [bases A | bases B | meta A | meta B]
                cap = rec_ws.numel() // 24 // 256 * 256
                meta = rec_ws[20 * cap: 20 * cap + 4 * n_records].view(torch.int32).clone()
                kept.append((plan_ws, merge_ws, meta, w1 - w0))
            rec_ws = None
            need = _lib.check(L.pg_mini_shuffle_bytes_merged(n_words, keep.n_rows, vsize, self.desc()))
            if self._shuffle_ws is None or self._shuffle_ws.numel() < need:
                self._shuffle_ws = None
                self._shuffle_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            sws = self._shuffle_ws
            _lib.check(L.pg_mini_lookup_begin(self.desc(), rows_arg(keep), n_words, vsize, sws.data_ptr(), sws.numel(), sp))
            for plan_ws, merge_ws, meta, nw in kept:
                _lib.check(L.pg_mini_lookup_piece(self.desc(), rows_arg(keep), plan_ws.data_ptr(), plan_ws.numel(), nw, meta.data_ptr(), n_words, window, vsize,
                                                  sws.data_ptr(), sws.numel(), merge_ws.data_ptr(), self.status.data_ptr(), sp))
            # (``abundance_from_records`` hands a plan workspace of the whole range to pg_mini_abundance_from_emitted, which only checks its size)
            whole = torch.empty(_lib.check(L.pg_mini_plan_bytes(n_words, self.desc())), dtype=torch.uint8, device=self.device)
        self._mini_plan = (("pieces", len(ranges)), whole, sum(int(m.numel()) for _, _, m, _ in kept), keep, (stream.codes, table_plane), 0)
        self._mini_pieces = len(ranges)
        self._mini_optimistic = None
        self._empty = False
        self._records = (keep, n_words)
        self._emitted = (window, vsize)
        if check:
            self.check_status()
        return self

    def plan_counts(self) -> tuple:
        """(records, records of more than four k-mers) of the partition plan in use -- read from the device on first use (a plan
        picked up without a host wait keeps them there)"""
        key, ws, n_records, keep, held, n_long = self._mini_plan
        if n_records is None:
            head = ws[:24].view(torch.int64).cpu()
            n_records, n_long = int(head[0]), int(head[2])
            self._mini_plan = (key, ws, n_records, keep, held, n_long)
        return n_records, n_long

    def lookup_half(self, bins: torch.Tensor, bin_elem: torch.Tensor) -> None:
        """N > 1 ranks: finish a ``count_half`` -- ``bins`` (int16 view of what the bucket owners sent back: bin + 1 of every entry in
        the merged table, in the order the entries were sent), ``bin_elem[b]`` = where bucket b's bins start -- lookups of the
        provisional words and the row-group scatter; ``features`` then reads the rows from the shuffled words"""
        fill, n_words, keep, window, vsize = self._half
        assert bins.dtype == torch.int16 and bin_elem.dtype == torch.int64 and bin_elem.numel() == self.n_buckets
        L = _lib.load()
        plan_ws = self._mini_plan[1]
        with torch.cuda.device(self.device):
            _lib.check(L.pg_mini_lookup_half(self.desc(), C.byref(keep.rows_desc), plan_ws.data_ptr(), plan_ws.numel(),
                                             self._mini_rec_ws.data_ptr(), self._mini_rec_ws.numel(), n_words, vsize,
                                             self._shuffle_ws.data_ptr(), self._shuffle_ws.numel(),
                                             self._merge_ws.data_ptr() if getattr(self, "_merge_ws", None) is not None else None,
                                             self._merge_ws.numel() if getattr(self, "_merge_ws", None) is not None else 0,
                                             self._half_ws.data_ptr(), self._half_ws.numel(),
                                             bins.data_ptr(), bin_elem.data_ptr(), self.status.data_ptr(), _stream_ptr(self.device)))
        self._records = (keep, n_words)
        self._emitted = (window, vsize)
        self._half = None

    def prefetch_plan(self, stream: ReadStream, rows: "Plan | None", side: "torch.cuda.Stream",
                      after: "torch.cuda.Event | None" = None) -> None:
        """compute the partition plan of the NEXT count of ``stream`` (whole range, strict validity) on the stream ``side``, into a
        workspace of its own: ``side`` waits for what the current stream has enqueued so far -- call this right after ``count`` and
        the plan of batch i + 1 runs under the row histograms and the encode of batch i instead of in front of its own count.
        The next ``count`` of the same stream and rows picks it up (and waits for it); any other count ignores it."""
        if self.kind not in ("mini", "miniw"):
            raise ValueError("prefetch_plan() is for mini tables")
        if stream.table_valid(False) is not stream.valid or not stream.rows_inside_table:
            return                                               # (soft-masked / quality-masked input: plan inside the count)
        L = _lib.load()
        keep = rows if (rows is not None and rows.shuffle_ok and rows.n_rows <= _lib.MINI_MAX_ROWS) else None
        n_words = stream.n_words
        key = self._plan_key(stream.codes, stream.valid, 0, n_words, keep, False, self.log2_slots, self.log2_bucket)
        need = _lib.check(L.pg_mini_plan_bytes(n_words, self.desc()))
        ws = self._mini_spare if self._mini_spare is not None and self._mini_spare.numel() == need else None
        self._mini_spare = None
        if ws is None:
            # a fresh workspace is allocated ON the side stream: a block the caching allocator hands out there is free of pending
            # work of the current stream (with ``after`` set the side stream does not wait for that stream's tail).  The spare
            # workspace was last read by the count BEFORE the one just enqueued, which every ``after`` event lies behind.
            with torch.cuda.stream(side):
                ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            ws.record_stream(torch.cuda.current_stream(self.device))         # (the count that consumes it runs there)
        # ``after`` = an event recorded before this batch's count was enqueued: the plan then runs BESIDE the count (its kernel
        # is small enough to share the CUs with it) instead of behind it
        if after == "first-pass":
            _lib.check(L.pg_mini_wait_first_pass(side.cuda_stream))   # behind the first scatter pass of the count just enqueued
        elif after is not None:
            side.wait_event(after)
        else:
            side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.device(self.device), torch.cuda.stream(side):
            _lib.check(L.pg_mini_plan(stream.codes.data_ptr(), stream.valid.data_ptr(), 0, n_words, self.desc(),
                                      C.byref(keep.rows_desc) if keep is not None else None, ws.data_ptr(), ws.numel(), side.cuda_stream))
            event = torch.cuda.Event()
            event.record(side)
        ws.record_stream(side)
        self._mini_next = (key, ws, event, keep, (stream.codes, stream.valid))

    def can_shuffle(self, plan: "Plan", window: int, vsize: int) -> bool:
        """can ``abundance_from_records`` build the rows of this plan (instead of table lookups)?"""
        if self.kind in ("mini", "miniw"):
            return self._records is not None and self._records[0] is plan and self._emitted == (int(window), int(vsize))
        return self.has_records_for(plan, vsize)

    def has_records_for(self, plan: "Plan", vsize: int) -> bool:
        return (self._records is not None and self._records[0] is plan and vsize <= _lib.SHUFFLE_MAX_VSIZE
                and self.kind == "hash" and bool(self.log2_bucket))

    def abundance_from_records(self, plan: "Plan", window: int, vsize: int, out: torch.Tensor) -> torch.Tensor:
        """abundance rows by shuffle: bucket-wise LDS lookups of the kept records + row-group scatter + LDS row histograms"""
        if not self.can_shuffle(plan, window, vsize):
            raise RuntimeError("no partition records for these rows: count(stream, rows=plan) first")
        n_words = self._records[1]
        L = _lib.load()
        if self.kind in ("mini", "miniw"):
            plan_ws = self._mini_plan[1]
            with torch.cuda.device(self.device):
                _lib.check(L.pg_mini_abundance_from_emitted(self.desc(), C.byref(plan.rows_desc), vsize, out.data_ptr(), plan_ws.data_ptr(),
                                                            plan_ws.numel(), n_words, self._shuffle_ws.data_ptr(), self._shuffle_ws.numel(),
                                                            _stream_ptr(self.device)))
            self._emitted = None        # the row shuffle reuses the emitted words' buffer
            self._records = None
            return out
        emitted = self._emitted == (int(window), int(vsize))          # the lookup pass already ran inside the count
        sws = self._shuffle_ws if emitted else self._shuffle_workspace_for(n_words, plan.n_rows, vsize)
        fn = L.pg_abundance_from_emitted if emitted else L.pg_abundance_from_records
        with torch.cuda.device(self.device):
            _lib.check(fn(self.desc(), C.byref(plan.rows_desc), window, vsize, out.data_ptr(),
                          self._workspace.data_ptr(), self._workspace.numel(), n_words, sws.data_ptr(), sws.numel(), _stream_ptr(self.device)))
        self._emitted = None            # the row shuffle reuses the emitted words' buffer: they are gone now
        return out

    def release_workspaces(self) -> None:
        """give the scratch of the counting pipelines back (record buffers, word buffers, plans: tens of GB at BASELINE sizes); the
        table keeps its counts, the next count allocates again"""
        self._workspace = self._shuffle_ws = self._mini_rec_ws = self._mini_spare = None
        self._merge_ws = self._half_ws = None
        self._mini_plan = self._mini_next = None
        self._mini_sized_for = self._mini_optimistic = None
        self._records = self._emitted = self._half = None
        if self.data.is_cuda:
            torch.cuda.empty_cache()

    def check_status(self) -> None:
        """raise what the kernels reported in the status word (include/pangaea_feat.h: PG_STATUS_*)"""
        if self.kind == "dense":
            return
        st = int(self.status[0].item())
        if st & _lib.STATUS_BOUNDS:
            raise RuntimeError("a kernel of the checked build was about to store outside its buffer (PG_STATUS_BOUNDS): the results are incomplete")
        if st & _lib.STATUS_PLAN_MISMATCH:
            again = getattr(self, "_mini_optimistic", None)
            if again is not None and self._mini_plan is not None:
                # the count ran on the previous batch's workspaces without waiting for its plan's record counts, and this batch has
                # more records than they hold: nothing was written -- read the counts, size the workspaces, count again
                self._mini_optimistic = None
                stream, word_begin, word_end, rows, emit, half = again
                self.status.zero_()
                self.plan_counts()
                self._empty = True
                if half:
                    self.count_half(stream, rows, emit, check=False)
                else:
                    self.count(stream, word_begin, word_end, check=False, rows=rows, emit=emit)
                return self.check_status()
            self._mini_plan = None
            raise RuntimeError("the partition plan did not describe this stream (PG_STATUS_PLAN_MISMATCH): nothing was counted")
        if st != 0:
            raise _lib.PangaeaError(_lib.PG_ETABLEFULL, f"hash table with 2^{self.log2_slots} slots is full")

    def merge(self, pairs: torch.Tensor, check: bool = True, pending_ok: bool = False) -> "KmerTable":
        """add (key << 22 | count) pairs (slot format, key = key42(code)), e.g. the compacted table of another GPU"""
        if not pending_ok:
            self._require_counts()
        if self.kind not in ("hash", "mini"):
            raise ValueError("merge() is for hash and mini tables; dense tables are summed with all_reduce")
        pairs = pairs.to(self.device, torch.int64).contiguous()
        if self._empty and self.log2_bucket:
            self.data.zero_()            # a reset() bucketed table is only logically empty
        self._empty = False
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_kmer_merge(pairs.data_ptr(), pairs.numel(), self.desc(), self.status.data_ptr(),
                                                 _stream_ptr(self.device)))
        if check:
            self.check_status()
        return self

    @property
    def n_buckets(self) -> int:
        return 1 << (self.log2_slots - self.log2_bucket) if self.kind in ("hash", "mini", "miniw") and self.log2_bucket else 1

    def bucket_counts(self) -> torch.Tensor:
        """occupied slots per bucket, int64 [n_buckets] -- the segment lengths of ``compact()`` (slot order = bucket order)"""
        self._require_counts()
        if self.kind != "hash":
            raise ValueError("bucket_counts() is for hash tables")
        return (self.data.view(self.n_buckets, -1) != 0).sum(dim=1)

    def merge_parts(self, parts, check: bool = True) -> "KmerTable":
        """add other tables of the SAME geometry, each given as (compact(), bucket_counts()); bucket by bucket inside
        LDS when the buckets are LDS-sized, else with global atomics"""
        self._require_counts()
        parts = [(p.to(self.device, torch.int64), c.to(self.device, torch.int64)) for p, c in parts if p.numel()]
        if not parts:
            return self
        if not (self.kind == "hash" and 0 < self.log2_bucket <= _lib.BUCKET_MAX_LOG2_SLOTS):
            for p, _ in parts:
                self.merge(p, check=False)
        else:
            if self._empty:
                self.data.zero_()
            self._empty = False
            pairs = torch.cat([p for p, _ in parts]).contiguous()
            seg, base = [], 0
            for p, c in parts:
                if c.numel() != self.n_buckets or int(c.sum().item()) != p.numel():
                    raise ValueError("merge_parts: bucket counts do not describe the pairs (different table geometry?)")
                seg.append(torch.cat([c.new_zeros(1), torch.cumsum(c, 0)]) + base)
                base += p.numel()
            seg = torch.stack(seg).contiguous()
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().pg_kmer_merge_bucketed(pairs.data_ptr(), seg.data_ptr(), len(parts), self.desc(),
                                                              self.status.data_ptr(), _stream_ptr(self.device)))
        if check:
            self.check_status()
        return self

    # ---- the exchange step's kernels (dist.exchange_table): fills -> bucket-ordered compaction -> rebuild from all parts

    def _bucketed(self) -> bool:
        return self.kind == "hash" and 0 < self.log2_bucket <= _lib.BUCKET_MAX_LOG2_SLOTS

    def bucket_fill(self) -> torch.Tensor:
        """``bucket_counts()`` by one kernel pass over the table (int64 [n_buckets], on the device)"""
        self._require_counts()
        if not self._bucketed():
            raise ValueError("bucket_fill() is for bucketed hash tables")
        if self._empty:
            self.data.zero_()
            self._empty = False
        fill = torch.empty(self.n_buckets, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_table_bucket_fill(self.desc(), fill.data_ptr(), _stream_ptr(self.device)))
        return fill

    def compact_into(self, out: torch.Tensor, seg: torch.Tensor) -> None:
        """occupied slots, bucket after bucket, into ``out`` at the offsets ``seg`` (int64 [n_buckets + 1], the exclusive
        scan of ``bucket_fill()``); order inside a bucket is unspecified"""
        self._require_counts()
        if not self._bucketed():
            raise ValueError("compact_into() is for bucketed hash tables")
        _require_gpu(out, "the output")
        assert out.dtype == torch.int64 and out.is_contiguous() and seg.dtype == torch.int64 and seg.numel() == self.n_buckets + 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_table_compact(self.desc(), seg.data_ptr(), out.data_ptr(), _stream_ptr(self.device)))

    def _require_counts(self) -> None:
        if self._deferred is not None:
            raise RuntimeError("this table was counted in deferred form and holds no counts until dist.exchange_table / rebuild_from")

    @property
    def pending(self) -> bool:
        """counted in deferred form: the slots hold nothing until ``rebuild_from`` (``dist.exchange_table``)"""
        return self._deferred is not None

    def deferred_fill(self) -> torch.Tensor:
        return self._deferred[0]

    def deferred_compact_into(self, out: torch.Tensor, seg: torch.Tensor) -> None:
        """the deferred count's entries, bucket after bucket, into ``out`` at the offsets ``seg`` (as ``compact_into``)"""
        fill, n_words = self._deferred
        _require_gpu(out, "the output")
        assert out.dtype == torch.int64 and out.is_contiguous() and seg.dtype == torch.int64 and seg.numel() == self.n_buckets + 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_deferred_gather(self.desc(), self._workspace.data_ptr(), self._workspace.numel(), n_words,
                                                      fill.data_ptr(), seg.data_ptr(), out.data_ptr(), _stream_ptr(self.device)))

    @property
    def tag_bits(self) -> int:
        """bits of a key below its bucket id (the exchange sends entries as such tags when they fit 31 bits)"""
        return 42 - (self.log2_slots - self.log2_bucket)

    def deferred_planes_into(self, out: torch.Tensor, tag_elem: torch.Tensor, cnt_elem: torch.Tensor,
                             overflow: torch.Tensor, overflow_count: torch.Tensor) -> None:
        """the deferred count's entries in the 6-byte exchange format: bucket b's tags from element ``tag_elem[b]`` of the
        uint32 view of ``out`` (a byte buffer), its counts from element ``cnt_elem[b]`` of the uint16 view; counts beyond
        0xffff leave their remainder as whole entries in ``overflow`` (``overflow_count``: int64 device counter)"""
        fill, n_words = self._deferred
        _require_gpu(out, "the output")
        assert out.dtype == torch.uint8 and out.is_contiguous() and out.data_ptr() % 16 == 0
        assert tag_elem.dtype == cnt_elem.dtype == torch.int64 and tag_elem.numel() == cnt_elem.numel() == self.n_buckets
        assert overflow.dtype == torch.int64 and overflow_count.dtype == torch.int64
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_deferred_gather_planes(self.desc(), self._workspace.data_ptr(), self._workspace.numel(), n_words,
                                                             fill.data_ptr(), tag_elem.data_ptr(), cnt_elem.data_ptr(), out.data_ptr(),
                                                             overflow.data_ptr(), overflow_count.data_ptr(), overflow.numel(),
                                                             self.status.data_ptr(), _stream_ptr(self.device)))

    def bucket_fill_range(self, buckets: tuple) -> torch.Tensor:
        """occupied slots of buckets ``[begin, end)`` (int64 [end - begin]); the rest of the table is not read"""
        b0, b1 = buckets
        fill = torch.empty(b1 - b0, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_table_bucket_fill_range(self.desc(), b0, b1, fill.data_ptr(), _stream_ptr(self.device)))
        return fill

    def compact_planes_range(self, buckets: tuple, out: torch.Tensor, tag_elem: torch.Tensor, cnt_elem: torch.Tensor,
                             overflow: torch.Tensor, overflow_count: torch.Tensor) -> None:
        """buckets ``[begin, end)`` of the table in the 6-byte exchange format (as ``deferred_planes_into``; the element
        indices are per bucket of the range)"""
        b0, b1 = buckets
        assert out.dtype == torch.uint8 and out.is_contiguous() and out.data_ptr() % 16 == 0
        assert tag_elem.dtype == cnt_elem.dtype == torch.int64 and tag_elem.numel() == cnt_elem.numel() == b1 - b0
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_table_compact_planes_range(self.desc(), b0, b1, tag_elem.data_ptr(), cnt_elem.data_ptr(), out.data_ptr(),
                                                                 overflow.data_ptr(), overflow_count.data_ptr(), overflow.numel(),
                                                                 self.status.data_ptr(), _stream_ptr(self.device)))

    def mark_rebuilt(self) -> None:
        """every bucket range has been rebuilt: the table holds counts again"""
        self._empty = False
        self._deferred = None

    def rebuild_from_planes(self, buf: torch.Tensor, part_stride: int, cap: int, seg: torch.Tensor, buckets: tuple,
                            in_order: bool = True) -> None:
        """rebuild buckets ``[begin, end)`` from gathered 6-byte planes (``seg`` int64 [n_parts, end - begin + 1], indices
        inside every part's range)"""
        b0, b1 = buckets
        _require_gpu(buf, "the gathered planes")
        assert buf.dtype == torch.uint8 and seg.dtype == torch.int64 and seg.is_contiguous() and seg.shape[1] == b1 - b0 + 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_kmer_rebuild_planes_range(buf.data_ptr(), int(part_stride), int(cap), seg.data_ptr(), int(seg.shape[0]),
                                                                self.desc(), b0, b1, self.status.data_ptr(), _stream_ptr(self.device)))
        if in_order and b1 == self.n_buckets:       # ranges rebuilt in another order: the caller calls mark_rebuilt()
            self.mark_rebuilt()

    def rebuild_from(self, pairs: torch.Tensor, seg: torch.Tensor, check: bool = True, buckets: tuple | None = None) -> "KmerTable":
        """replace the table by the merge of ``seg.shape[0]`` bucket-ordered compacted tables of this geometry laid out in
        ``pairs`` (``seg`` int64 [n_parts, n_buckets + 1], absolute offsets): one workgroup per bucket, inside LDS.
        ``buckets`` = (begin, end) rebuilds that bucket range only (``seg`` then [n_parts, end - begin + 1]); the table
        holds counts again once every range has been rebuilt -- the caller's business."""
        if not self._bucketed():
            raise ValueError("rebuild_from() is for bucketed hash tables")
        _require_gpu(pairs, "the pairs")
        b0, b1 = buckets if buckets is not None else (0, self.n_buckets)
        assert pairs.dtype == torch.int64 and seg.dtype == torch.int64 and seg.is_contiguous() and seg.shape[1] == b1 - b0 + 1
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pg_kmer_rebuild_bucketed_range(pairs.data_ptr(), seg.data_ptr(), int(seg.shape[0]), self.desc(), b0, b1,
                                                                  self.status.data_ptr(), _stream_ptr(self.device)))
        if buckets is None or b1 == self.n_buckets:
            self._empty = False             # (the row-tagged records of this rank's count stay valid: the geometry is the same)
            self._deferred = None
        if check:
            self.check_status()
        return self

    def compact(self) -> torch.Tensor:
        """occupied slots of a hash table as an int64 vector (slot format)"""
        self._require_counts()
        if self.kind not in ("hash", "mini"):
            raise ValueError("compact() is for hash and mini tables")
        step = 1 << 30                       # torch's masked select overflows its 32-bit indexing at 2^31 elements
        if self.data.numel() <= step:
            return self.data[self.data != 0]
        return torch.cat([c[c != 0] for c in self.data.split(step)])

    def occupancy(self) -> float:
        self._require_counts()
        if self.kind in ("wide", "miniw"):
            return float(torch.count_nonzero(self._wide_parts()[0]).item()) / (1 << self.log2_slots)
        if self.kind not in ("hash", "mini"):
            return float("nan")
        return float(torch.count_nonzero(self.data).item()) / self.data.numel()

    def items(self):
        """(codes uint64, counts uint64) sorted by code -- host copies, for tests"""
        self._require_counts()
        if self.kind == "dense":
            t = self.data.cpu().numpy().view(np.uint32)
            codes = np.nonzero(t)[0].astype(np.uint64)
            return codes, t[codes.astype(np.int64)].astype(np.uint64)
        if self.kind in ("wide", "miniw"):
            keys, cnts = self._wide_parts()
            occ = keys != 0
            codes = (keys[occ] - 1).cpu().numpy().view(np.uint64)
            counts = cnts[occ].cpu().numpy().view(np.uint32).astype(np.uint64)
            order = np.argsort(codes)
            return codes[order], counts[order]
        s = self.compact().cpu().numpy().view(np.uint64)
        if self.kind == "mini":
            codes, counts = s >> np.uint64(_lib.HASH_COUNT_BITS), s & np.uint64((1 << _lib.HASH_COUNT_BITS) - 1)
            order = np.argsort(codes)
            return codes[order], counts[order]
        codes, counts = key42_inverse(s >> np.uint64(_lib.HASH_COUNT_BITS)), s & np.uint64((1 << _lib.HASH_COUNT_BITS) - 1)
        order = np.argsort(codes)
        return codes[order], counts[order]


_KEY_MASK = np.uint64((1 << 42) - 1)


def key42(codes: np.ndarray) -> np.ndarray:
    """the slot key of canonical codes (uint64 < 2^42): the library's pg_key42, a bijection on 42 bits"""
    x = np.asarray(codes, dtype=np.uint64).copy()
    for m in (_lib.KEY42_M1, _lib.KEY42_M2):
        x ^= x >> np.uint64(21)
        x = (x * np.uint64(m)) & _KEY_MASK
    x ^= x >> np.uint64(21)
    return x


def key42_inverse(keys: np.ndarray) -> np.ndarray:
    """canonical codes of slot keys: the xorshifts are involutions on 42 bits, the odd multipliers have inverses mod 2^42"""
    x = np.asarray(keys, dtype=np.uint64).copy()
    for m in (_lib.KEY42_M2, _lib.KEY42_M1):
        x ^= x >> np.uint64(21)
        x = (x * np.uint64(pow(m, -1, 1 << 42))) & _KEY_MASK
    x ^= x >> np.uint64(21)
    return x


def distinct_sketch(stream: ReadStream, k: int, word_begin: int = 0, word_end: int | None = None,
                    lowercase_is_base: bool = False) -> torch.Tensor:
    """HyperLogLog registers (int32 [4096], on the device) of the stream's canonical k-mers.  The elementwise maximum of
    two sketches is the sketch of the union -- how the ranks of a multi-GPU job size their common table."""
    _require_gpu(stream.codes, "the read stream")
    dev = stream.device
    regs = torch.zeros(_lib.HLL_REGISTERS, dtype=torch.int32, device=dev)
    word_end = stream.n_words if word_end is None else word_end
    with torch.cuda.device(dev):
        valid = stream.table_valid(lowercase_is_base)
        _lib.check(_lib.load().pg_kmer_distinct_sketch(stream.codes.data_ptr(), valid.data_ptr(), word_begin, word_end, k,
                                                       regs.data_ptr(), _stream_ptr(dev)))
    return regs


def sketch_estimate(regs: torch.Tensor) -> int:
    """cardinality estimate of a HyperLogLog sketch (~1.6 % standard error at 4096 registers)"""
    r = regs.cpu().numpy().astype(np.float64)
    m = float(len(r))
    est = (0.7213 / (1.0 + 1.079 / m)) * m * m / np.sum(np.exp2(-r))
    zeros = int((r == 0).sum())
    if est <= 2.5 * m and zeros:
        est = m * math.log(m / zeros)                      # linear counting for small cardinalities
    return int(est)


def estimate_distinct(stream: ReadStream, k: int, word_begin: int = 0, word_end: int | None = None,
                      lowercase_is_base: bool = False) -> int:
    """HyperLogLog estimate of the number of distinct canonical k-mers"""
    return sketch_estimate(distinct_sketch(stream, k, word_begin, word_end, lowercase_is_base))


def count_kmers(stream: ReadStream, k: int, kind: str | None = None, distinct_hint: int | None = None,
                max_log2_slots: int = 36, log2_bucket: int | None = None, rows: "Plan | None" = None,
                emit: tuple | None = None, lowercase_is_base: bool = False, load: float | None = None) -> KmerTable:
    """build the table of one stream; a full hash table is re-built with four times the slots.  ``emit`` = (window,
    vector_size) fuses the lookup pass of the abundance rows into the count where that applies (``KmerTable.count``)."""
    resolved = kind or KmerTable.default_kind(k)
    auto_mini = (kind is None and resolved in ("hash", "wide") and rows is not None and emit is not None and rows.shuffle_ok
                 and rows.n_rows <= _lib.MINI_MAX_ROWS and 1 <= emit[1] <= _lib.SHUFFLE_MAX_VSIZE and emit[0] >= 1
                 and (resolved == "wide" or emit[0] * emit[1] <= _lib.HASH_COUNT_SAT) and _lib.MINI_MIN_K <= k <= _lib.WIDE_MAX_K
                 and os.environ.get("PANGAEA_NO_MINI", "0") in ("", "0"))
    if distinct_hint is None and resolved != "dense":
        # size from a HyperLogLog pass (as cheap as the bucket histogram) instead of guessing the coverage; +10 % covers
        # the estimator's error, load 0.4 leaves room for per-bucket variance
        distinct_hint = max(1 << 13, int(1.1 * estimate_distinct(stream, k, lowercase_is_base=lowercase_is_base)))
        load = 0.4 if load is None else load
    elif load is None:
        load = 0.5
    if auto_mini and log2_bucket is None:
        # one GPU, rows and abundance parameters known: the super-k-mer pipeline (table by minimizer buckets) where its geometry
        # (at most 2^15 buckets of 2^14 slots) holds the table
        want = max(1024, int(distinct_hint / load))
        log2 = max(10, math.ceil(math.log2(want)))
        if KmerTable.mini_applies(k, log2):
            kind = "mini" if k <= _lib.HASH_MAX_K else "miniw"
        elif KmerTable.mini_applies(k, log2 - 1) and distinct_hint <= 0.7 * (1 << (log2 - 1)):
            # the largest geometry (2^16 buckets) at a higher load still beats the other pipelines by far (k = 31, 10 M pairs:
            # 2^29 slots at load 0.45 instead of a 2^30-slot direct table)
            kind = "mini" if k <= _lib.HASH_MAX_K else "miniw"
            load = distinct_hint / float(1 << (log2 - 1))
    table = KmerTable.alloc(k, stream.device, kind, distinct_hint, load=load, log2_bucket=log2_bucket)
    while True:
        try:
            return table.count(stream, rows=rows, emit=emit, lowercase_is_base=lowercase_is_base)
        except _lib.PangaeaError as e:
            if e.code != _lib.PG_ETABLEFULL or table.log2_slots >= max_log2_slots:
                raise
            log2 = min(max_log2_slots, table.log2_slots + 2)
            if table.kind in ("mini", "miniw"):
                lb = min(KmerTable.mini_max_log2_bucket(k), table.log2_bucket + 2)
                del table
                table = (KmerTable.mini_with_slots(k, stream.device, log2, lb) if KmerTable.mini_applies(k, log2, lb)
                         else KmerTable.with_slots(k, stream.device, log2) if k <= _lib.HASH_MAX_K
                         else KmerTable.wide_with_slots(k, stream.device, log2))
                continue
            if table.kind == "wide":
                del table
                table = KmerTable.wide_with_slots(k, stream.device, log2)
                continue
            lb = None if log2_bucket is None else min(log2_bucket + 2, _lib.BUCKET_MAX_LOG2_SLOTS)
            if lb is not None and log2 - lb > _lib.BUCKET_MAX_LOG2_BUCKETS:
                lb = 0
            del table
            table = KmerTable.with_slots(k, stream.device, log2, lb)


# ---------------------------------------------------------------------------------------- workspaces ahead of the data

def prewarm_workspaces(device, n_pairs: int, k: int, vsize: int, read_len: int = 150):
    """Start the allocation of the scratch buffers that the super-k-mer pipeline will ask for, on a helper thread, and return the
    thread (``join()`` it before counting; a data set that turns out larger simply allocates again).

    What a FIRST pass over a data set costs besides its kernels is mostly ``hipMalloc``: record buffers, slot buffer and the row
    shuffle's words are tens of GB at BASELINE sizes (38 GB per 10 M read pairs), their sizes follow from the number of read pairs
    within a few percent, and the ingest -- host threads parsing FASTQ -- leaves the GPU's driver idle meanwhile.  The buffers
    are allocated through torch's caching allocator and given straight back to it: the later requests (same thread or not, same
    stream) are served from those blocks (a larger block is split).  ``pangaea.py`` extracts features once per data set
    (/root/reference/src/pangaea.py:70), so the first pass IS the user's pass.  Only for 13 <= k <= 31 (the pipeline that
    needs the buffers); PANGAEA_PREWARM=0 turns it off."""
    import threading
    if not (_lib.MINI_MIN_K <= k <= _lib.WIDE_MAX_K) or n_pairs <= 0 or os.environ.get("PANGAEA_PREWARM", "1") in ("", "0"):
        return None
    device = torch.device(device)
    if device.type != "cuda":
        return None
    L = _lib.load()
    n_words = (n_pairs * 2 * (read_len + 1) + 31) // 32
    n_words = (n_words + _lib.WORD_ALIGN - 1) // _lib.WORD_ALIGN * _lib.WORD_ALIGN
    m = 13 if k >= 16 else 11                        # (pg_device.hpp: mini_m)
    w = min(k - m + 1, 9)
    records = int(n_words * (32.0 / ((w + 1) / 2.0) + 1.0) * 0.9)     # (k = 21: 6.7 per word; measured 6.45)
    kind = _lib.TABLE_MINI if k <= _lib.HASH_MAX_K else _lib.TABLE_MINI_WIDE
    lb = _lib.BUCKET_MAX_LOG2_SLOTS if k <= _lib.HASH_MAX_K else _lib.MINI_WIDE_MAX_LOG2_BUCKET_SLOTS
    desc = _lib.pg_table(kind, k, lb + 15, lb, None)
    n_rows = max(1, n_pairs // 100)
    sizes = []
    try:
        sizes.append(_lib.check(L.pg_mini_records_bytes(records, C.byref(desc))))
        sizes.append(_lib.check(L.pg_mini_shuffle_bytes_merged(n_words, n_rows, vsize, C.byref(desc))))
        sizes.append(4 * _lib.check(L.pg_mini_merge_words(n_words, records, records // 2, C.byref(desc))))
    except _lib.PangaeaError:
        return None

    def work():
        try:
            with torch.cuda.device(device):
                held = [torch.empty(int(b), dtype=torch.uint8, device=device) for b in sizes]
                del held
        except RuntimeError:                         # (out of memory: the pipeline will say so itself, with its real sizes)
            pass

    th = threading.Thread(target=work, name="pangaea-prewarm", daemon=True)
    th.start()
    return th


# ---------------------------------------------------------------------------------------- TNF columns

_COLMAP_CACHE: dict = {}


def tnf_ncols(k: int) -> int:
    return _lib.check(_lib.load().pg_tnf_ncols(k))


def tnf_colmap(k: int, device=None):
    """(colmap int16-as-uint16 tensor [4^k], column codes uint32 [ncols])"""
    key = (k, str(device))
    if key not in _COLMAP_CACHE:
        n = tnf_ncols(k)
        colmap = np.zeros(4 ** k, dtype=np.uint16)
        codes = np.zeros(n, dtype=np.uint32)
        _lib.check(_lib.load().pg_tnf_colmap(k, colmap.ctypes.data, codes.ctypes.data))
        t = torch.from_numpy(colmap.view(np.int16))
        _COLMAP_CACHE[key] = (t.to(device) if device is not None else t, codes)
    return _COLMAP_CACHE[key]


# ---------------------------------------------------------------------------------------- rows


def plan_segments(rows: Rows, seg_chars: int = DEFAULT_SEG_CHARS):
    L = _lib.load()
    start = np.ascontiguousarray(rows.start, dtype=np.int64)
    end = np.ascontiguousarray(rows.end, dtype=np.int64)
    n = _lib.check(L.pg_plan_segments(start.ctypes.data, end.ctypes.data, len(start), seg_chars, None, None, None))
    seg_row = np.zeros(n, dtype=np.int32)
    seg_start = np.zeros(n, dtype=np.int64)
    seg_end = np.zeros(n, dtype=np.int64)
    _lib.check(L.pg_plan_segments(start.ctypes.data, end.ctypes.data, len(start), seg_chars, seg_row.ctypes.data,
                                  seg_start.ctypes.data, seg_end.ctypes.data))
    return seg_row, seg_start, seg_end


class Plan:
    """device copy of a row set: its work segments (lookup kernels) and its row ranges (row ids in the partition
    records of the shuffle path); re-usable across launches"""

    def __init__(self, rows: Rows, device, seg_chars: int = DEFAULT_SEG_CHARS):
        r, s, e = plan_segments(rows, seg_chars)
        self.n_rows, self.n_segs = len(rows), len(r)
        self.seg_row = torch.from_numpy(r).to(device)
        self.seg_start = torch.from_numpy(s).to(device)
        self.seg_end = torch.from_numpy(e).to(device)
        start = np.ascontiguousarray(rows.start, dtype=np.int64)
        end = np.ascontiguousarray(rows.end, dtype=np.int64)
        # the shuffle path needs sorted, disjoint, non-empty rows (barcode runs are) and at most 2^22 - 2 of them
        self.shuffle_ok = bool(len(start) and len(start) <= _lib.MAX_ROWS and (end > start).all()
                               and (start[1:] >= end[:-1]).all())
        self.row_start = torch.from_numpy(start).to(device)
        self.row_end = torch.from_numpy(end).to(device)
        self.rows_desc = _lib.pg_rows(self.row_start.data_ptr(), self.row_end.data_ptr(), self.n_rows)


def features(stream: ReadStream, rows: Rows | Plan, k_tnf: int | None = 4, table: KmerTable | None = None,
             window: int = 10, vsize: int = 400, seg_chars: int = DEFAULT_SEG_CHARS,
             out_tnf: torch.Tensor | None = None, out_abd: torch.Tensor | None = None):
    """(tnf int32 [N, D] or None, abd int32 [N, V] or None) on the stream's device.

    tnf[r, c]  = occurrences of the c-th canonical k_tnf-mer in run r            (count_tnf.cpp:78-113)
    abd[r, b]  = k-mer occurrences of run r whose global multiplicity // window == b < vsize
                                                                                 (count_kmer.cpp:55-108)
    """
    _require_gpu(stream.codes, "the read stream")
    dev = stream.device
    plan = rows if isinstance(rows, Plan) else Plan(rows, dev, seg_chars)
    n = plan.n_rows
    tnf = abd = None
    colmap_ptr = None
    if k_tnf:
        colmap, _ = tnf_colmap(k_tnf, dev)
        colmap_ptr = colmap.data_ptr()
        tnf = out_tnf.zero_() if out_tnf is not None else torch.zeros((n, tnf_ncols(k_tnf)), dtype=torch.int32, device=dev)
    shuffle = table is not None and table.can_shuffle(plan, window, vsize)
    if table is not None:
        table._require_counts()
        if table.device != dev:
            raise ValueError("stream and table are on different devices")
        if shuffle:                 # every row is overwritten: no zero fill
            abd = out_abd if out_abd is not None else torch.empty((n, vsize), dtype=torch.int32, device=dev)
        else:
            abd = out_abd.zero_() if out_abd is not None else torch.zeros((n, vsize), dtype=torch.int32, device=dev)
    if tnf is None and abd is None:
        raise ValueError("nothing to compute: give k_tnf and/or a table")
    if plan.n_segs == 0:            # no rows (or only empty ones): the zero-filled matrices are the answer
        return tnf, abd
    if shuffle:
        table.abundance_from_records(plan, window, vsize, abd)
        if tnf is None:
            return tnf, abd
        table = None                         # the lookup kernel below only counts TNF
    with torch.cuda.device(dev):
        _lib.check(_lib.load().pg_features(
            stream.codes.data_ptr(), stream.valid.data_ptr(), stream.n_words,
            plan.seg_row.data_ptr(), plan.seg_start.data_ptr(), plan.seg_end.data_ptr(), plan.n_segs,
            k_tnf or 0, colmap_ptr, tnf.data_ptr() if tnf is not None else None,
            table.desc() if table is not None else None, window, vsize,
            abd.data_ptr() if (abd is not None and table is not None) else None, _stream_ptr(dev)))
    return tnf, abd
