"""The packed read stream and its barcode runs (layout: include/pangaea_feat.h).

``ReadStream`` is what the reference's two producer loops hand to their worker threads
(count_tnf.cpp:238-289, count_kmer.cpp:239-281), kept for the whole file at once: 2-bit codes + 1-bit
validity for every character of ``read1 N read2 N ...`` in file order, plus the character offsets of the runs.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib


def words_for(n_chars: int) -> int:
    return int(_lib.check(_lib.load().pg_words_for(int(n_chars))))


@dataclass
class Rows:
    """surviving runs: non-empty barcode and more than ``min_len`` characters (count_tnf.cpp:81)"""
    run_index: np.ndarray          # int64 [N]
    names: list                    # barcode of each row
    start: np.ndarray              # int64 [N] character range in the stream
    end: np.ndarray

    def __len__(self) -> int:
        return len(self.names)


class _IngestHandle:
    """owns a pg_reads handle: the stream arrays of a host ReadStream are views of the library's memory"""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.pg_reads_free(self._h)
            self._h = None


@dataclass
class ReadStream:
    codes: torch.Tensor            # int64 [n_words]: 32 characters per word, 2 bits each (A0 C1 T2 G3)
    valid: torch.Tensor            # int32 [n_words]: bit j set iff character j is one of ACGT
    n_chars: int
    run_off: np.ndarray            # int64 [n_runs + 1]
    run_names: list
    n_pairs: int = 0
    n_unpaired: int = 0
    mode: str = ""
    _owner: object = field(default=None, repr=False)     # keeps library-owned host memory alive
    # int32 [n_words] or None: bit j set iff character j is a LOWER-case a c g t.  jellyfish counts those as bases, the
    # reference's own counters reset on them: `valid` excludes them (their 2-bit codes are in `codes` all the same) and
    # ``KmerTable.count(..., lowercase_is_base=True)`` counts with ``valid | valid_lower``.  None: the input has none.
    valid_lower: torch.Tensor | None = None
    # int32 [n_words] or None: bit j set iff character j is a base (either case) with a quality character below '?'.  Only
    # paired (-1/-2) input has it: there the reference runs jellyfish with --min-qual-char=? (feature.py:76-83), which reads
    # such bases as N, while its own counters never look at qualities.  The TABLE is counted with ``table_valid()``, rows
    # with ``valid``.
    valid_lowq: torch.Tensor | None = None

    @property
    def n_words(self) -> int:
        return int(self.codes.numel())

    @property
    def device(self) -> torch.device:
        return self.codes.device

    def lenient_valid(self) -> torch.Tensor:
        """the validity plane under jellyfish's rule (lower-case bases are bases); ``valid`` itself when there are none"""
        if self.valid_lower is None:
            return self.valid
        if getattr(self, "_lenient", None) is None:
            self._lenient = self.valid | self.valid_lower
        return self._lenient

    def table_valid(self, lowercase_is_base: bool = True) -> torch.Tensor:
        """the validity plane the multiplicity table is counted with -- jellyfish's view of the reads: lower-case bases count
        (``lowercase_is_base``), bases below the quality threshold of paired input do not"""
        v = self.lenient_valid() if lowercase_is_base else self.valid
        if self.valid_lowq is None:
            return v
        key = "_table_valid_lc" if lowercase_is_base else "_table_valid"
        if getattr(self, key, None) is None:
            setattr(self, key, v & ~self.valid_lowq)
        return getattr(self, key)

    @property
    def rows_inside_table(self) -> bool:
        """is every k-mer of a row also a k-mer of the table's view?  (not when qualities mask bases: then the abundance
        rows are built by table lookups, the only form in which a row's k-mer may be absent from the table)"""
        return self.valid_lowq is None

    # ------------------------------------------------------------------ constructors

    @classmethod
    def from_fastq(cls, reads1: str, reads2: str | None = None, device: str | torch.device = "cpu") -> "ReadStream":
        """ingest an interleaved FASTQ (``reads2`` None) or an R1/R2 pair; gzip or plain"""
        L = _lib.load()
        if torch.device(device).type == "cuda":
            s = (cls._ingest_to_device(L, str(reads1), 0, 1, None, torch.device(device)) if reads2 is None
                 else cls._ingest_pair_to_device(L, str(reads1), str(reads2), torch.device(device)))
            if s is not None:
                return s
        h = C.c_void_p()
        _lib.check(L.pg_ingest_fastq(str(reads1).encode(), str(reads2).encode() if reads2 else None, C.byref(h)))
        return cls._from_handle(L, h).to(device)

    @classmethod
    def from_fastq_shard(cls, reads: str, part: int, n_parts: int, newlines_before=None,
                         device: str | torch.device = "cpu") -> "ReadStream":
        """shard ``part`` of ``n_parts`` of an uncompressed interleaved FASTQ: only that byte range is read and parsed,
        cut at run boundaries, so the shards' runs in rank order are the runs of the whole file.  ``newlines_before``
        [n_parts + 1] are the newline counts in front of every byte boundary (``count_newlines`` per range, prefix
        summed; ``pangaea_amd.dist.ingest_shard`` exchanges them between ranks); None counts them all here."""
        L = _lib.load()
        if newlines_before is None:
            newlines_before = np.concatenate([[0], np.cumsum([cls.count_newlines(reads, i, n_parts) for i in range(n_parts)])])
        before = np.ascontiguousarray(newlines_before, dtype=np.int64)
        assert before.shape == (n_parts + 1,)
        if torch.device(device).type == "cuda":
            s = cls._ingest_to_device(L, str(reads), int(part), int(n_parts), before, torch.device(device))
            if s is not None:
                return s
        h = C.c_void_p()
        _lib.check(L.pg_ingest_fastq_shard(str(reads).encode(), int(part), int(n_parts),
                                           before.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(h)))
        return cls._from_handle(L, h).to(device)

    @staticmethod
    def count_newlines(reads: str, part: int, n_parts: int) -> int:
        """newlines in byte range ``part`` of ``n_parts`` of an uncompressed file (RuntimeError for gzip input)"""
        n = C.c_int64()
        _lib.check(_lib.load().pg_fastq_count_newlines(str(reads).encode(), int(part), int(n_parts), C.byref(n)))
        return int(n.value)

    @classmethod
    def _ingest_to_device(cls, L, reads: str, part: int, n_parts: int, before, device) -> "ReadStream | None":
        """``pg_ingest_fastq_device``: the parser threads copy finished pieces to the GPU while the others parse on, the shift
        into place is a kernel, no host copy of the stream exists.  gzip input is inflated into an in-memory file first.  None
        when the input cannot go this way (a gzip shard, a gzip file whose text does not fit in memory, or
        ``PANGAEA_INGEST_ON_HOST=1``): the caller takes the host ingest and copies."""
        if os.environ.get("PANGAEA_INGEST_ON_HOST", "0") not in ("", "0"):
            return None
        memfd = -1
        try:
            with open(reads, "rb") as f:
                gz = f.read(2) == b"\x1f\x8b"
            size = os.path.getsize(reads)
        except OSError:
            return None                                      # (the host ingest reports it)
        if gz:
            # one inflate stream into an in-memory file (what pg_ingest_fastq does for its threaded parse), then the same
            # pieces, copies and placement as for a plain file; a shard of a gzip file would inflate all of it on every rank
            if n_parts > 1:
                return None
            fd, text = C.c_int(-1), C.c_int64(0)
            _lib.check(L.pg_inflate_to_memfd(reads.encode(), C.byref(fd), C.byref(text)))
            if fd.value < 0:
                return None                                  # (does not fit in memory: the host ingest streams it)
            memfd, size, reads = fd.value, int(text.value), f"/proc/self/fd/{fd.value}"
        try:
            return cls._ingest_plain_to_device(L, reads, part, n_parts, before, device, size)
        finally:
            if memfd >= 0:
                os.close(memfd)

    @classmethod
    def _ingest_plain_to_device(cls, L, reads: str, part: int, n_parts: int, before, device, size: int) -> "ReadStream | None":
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(device):
            cap = int(L.pg_ingest_staging_words(size))
            sc = torch.empty(cap, dtype=torch.int64, device=device)
            sv = torch.empty(cap, dtype=torch.int32, device=device)
            torch.cuda.current_stream().synchronize()        # (the copies run on the library's own streams)
            h = C.c_void_p()
            _lib.check(L.pg_ingest_fastq_device(reads.encode(), part, n_parts,
                                                None if before is None else before.ctypes.data_as(C.POINTER(C.c_int64)), size,
                                                C.c_void_p(sc.data_ptr()), C.c_void_p(sv.data_ptr()), cap, C.byref(h)))
            if not h:
                return None
            owner = _IngestHandle(L, h)
            nw, nr = L.pg_reads_n_words(h), L.pg_reads_n_runs(h)
            codes = torch.empty(nw, dtype=torch.int64, device=device)
            valid = torch.empty(nw, dtype=torch.int32, device=device)
            _lib.check(L.pg_ingest_place(h, C.c_void_p(sc.data_ptr()), C.c_void_p(sv.data_ptr()), cap,
                                         C.c_void_p(codes.data_ptr()), C.c_void_p(valid.data_ptr()), nw,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            lower_ptr = L.pg_reads_lower(h)
            lower = torch.from_numpy(np.ctypeslib.as_array(C.cast(lower_ptr, C.POINTER(C.c_int32)), shape=(nw,)).copy()).to(device) if lower_ptr else None
            run_off = np.ctypeslib.as_array(C.cast(L.pg_reads_run_off(h), C.POINTER(C.c_int64)), shape=(nr + 1,)).copy()
            names = cls._run_names(L, h, nr)
            out = cls(codes, valid, int(L.pg_reads_n_chars(h)), run_off, names, int(L.pg_reads_n_pairs(h)), int(L.pg_reads_n_unpaired(h)),
                      L.pg_reads_mode(h).decode(), valid_lower=lower)
            del owner
        return out

    @staticmethod
    def _plain_or_inflated(L, path: str):
        """(path to read, size of the text, descriptor to close or -1); None: a gzip file whose text may not be parked in memory"""
        with open(path, "rb") as f:
            gz = f.read(2) == b"\x1f\x8b"
        if not gz:
            return path, os.path.getsize(path), -1
        fd, text = C.c_int(-1), C.c_int64(0)
        _lib.check(L.pg_inflate_to_memfd(path.encode(), C.byref(fd), C.byref(text)))
        if fd.value < 0:
            return None
        return f"/proc/self/fd/{fd.value}", int(text.value), fd.value

    @classmethod
    def _ingest_pair_to_device(cls, L, reads1: str, reads2: str, device) -> "ReadStream | None":
        """``pg_ingest_fastq_pair_device``: -1 / -2 input with the copy to the GPU inside -- the threads pair and pack the records
        of a piece of R1 (and the same records of R2) and copy its streams (codes, validity, low quality) to the GPU while the
        others go on; placement is a kernel.  gzip files are inflated into in-memory files first, side by side.  None when the
        input cannot go this way (tiny files, a text that does not fit in memory, ``PANGAEA_INGEST_ON_HOST=1``)."""
        if os.environ.get("PANGAEA_INGEST_ON_HOST", "0") not in ("", "0"):
            return None
        opened = []
        try:
            try:
                for p in (reads1, reads2):
                    got = cls._plain_or_inflated(L, p)
                    if got is None:
                        return None
                    opened.append(got)
            except OSError:
                return None                                  # (the host ingest reports it)
            (p1, n1, _), (p2, n2, _) = opened
            if device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            with torch.cuda.device(device):
                cap = int(L.pg_ingest_pair_staging_words(n1, n2))
                sc = torch.empty(cap, dtype=torch.int64, device=device)
                sv = torch.empty(cap, dtype=torch.int32, device=device)
                sq = torch.empty(cap, dtype=torch.int32, device=device)
                torch.cuda.current_stream().synchronize()    # (the copies run on the library's own streams)
                h = C.c_void_p()
                _lib.check(L.pg_ingest_fastq_pair_device(p1.encode(), p2.encode(), n1, n2, C.c_void_p(sc.data_ptr()), C.c_void_p(sv.data_ptr()),
                                                         C.c_void_p(sq.data_ptr()), cap, C.byref(h)))
                if not h:
                    return None
                owner = _IngestHandle(L, h)
                nw, nr = L.pg_reads_n_words(h), L.pg_reads_n_runs(h)
                codes = torch.empty(nw, dtype=torch.int64, device=device)
                valid = torch.empty(nw, dtype=torch.int32, device=device)
                lowq = torch.empty(nw, dtype=torch.int32, device=device) if L.pg_reads_staged_lowq(h) else None
                _lib.check(L.pg_ingest_place_pair(h, C.c_void_p(sc.data_ptr()), C.c_void_p(sv.data_ptr()), C.c_void_p(sq.data_ptr()), cap,
                                                  C.c_void_p(codes.data_ptr()), C.c_void_p(valid.data_ptr()),
                                                  C.c_void_p(lowq.data_ptr()) if lowq is not None else None, nw,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))
                lower_ptr = L.pg_reads_lower(h)
                lower = torch.from_numpy(np.ctypeslib.as_array(C.cast(lower_ptr, C.POINTER(C.c_int32)), shape=(nw,)).copy()).to(device) if lower_ptr else None
                run_off = np.ctypeslib.as_array(C.cast(L.pg_reads_run_off(h), C.POINTER(C.c_int64)), shape=(nr + 1,)).copy()
                names = cls._run_names(L, h, nr)
                out = cls(codes, valid, int(L.pg_reads_n_chars(h)), run_off, names, int(L.pg_reads_n_pairs(h)), int(L.pg_reads_n_unpaired(h)),
                          L.pg_reads_mode(h).decode(), valid_lower=lower, valid_lowq=lowq)
                del owner
            return out
        finally:
            for _, _, fd in opened:
                if fd >= 0:
                    os.close(fd)

    @staticmethod
    def _run_names(L, h, nr: int) -> list:
        """the run names of a handle, fetched in one call"""
        need = int(L.pg_reads_run_names(h, None, 0))
        buf = C.create_string_buffer(max(1, need))
        L.pg_reads_run_names(h, buf, need)
        names = buf.raw[:need].decode().split("\0")[:-1] if need else []
        assert len(names) == nr
        return names

    @classmethod
    def _from_handle(cls, L, h) -> "ReadStream":
        owner = _IngestHandle(L, h)
        nw, nr = L.pg_reads_n_words(h), L.pg_reads_n_runs(h)
        # zero-copy views of the library's arrays (the handle is freed with the last reference to this stream)
        codes = np.ctypeslib.as_array(C.cast(L.pg_reads_codes(h), C.POINTER(C.c_int64)), shape=(nw,))
        valid = np.ctypeslib.as_array(C.cast(L.pg_reads_valid(h), C.POINTER(C.c_int32)), shape=(nw,))
        lower_ptr = L.pg_reads_lower(h)
        lower = torch.from_numpy(np.ctypeslib.as_array(C.cast(lower_ptr, C.POINTER(C.c_int32)), shape=(nw,))) if lower_ptr else None
        lowq_ptr = L.pg_reads_lowq(h)
        lowq = torch.from_numpy(np.ctypeslib.as_array(C.cast(lowq_ptr, C.POINTER(C.c_int32)), shape=(nw,))) if lowq_ptr else None
        run_off = np.ctypeslib.as_array(C.cast(L.pg_reads_run_off(h), C.POINTER(C.c_int64)), shape=(nr + 1,)).copy()
        names = cls._run_names(L, h, nr)
        out = cls(torch.from_numpy(codes), torch.from_numpy(valid), int(L.pg_reads_n_chars(h)), run_off, names,
                  int(L.pg_reads_n_pairs(h)), int(L.pg_reads_n_unpaired(h)), L.pg_reads_mode(h).decode(), _owner=owner,
                  valid_lower=lower, valid_lowq=lowq)
        return out

    @classmethod
    def from_runs(cls, runs, device: str | torch.device = "cpu") -> "ReadStream":
        """``runs`` = [(barcode, text)] with text already in run form (reads each followed by a non-base)"""
        L = _lib.load()
        text = b"".join(t if isinstance(t, bytes) else t.encode() for _, t in runs)
        off = np.zeros(len(runs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(t) for _, t in runs])
        nw = words_for(len(text))
        codes = np.zeros(nw, dtype=np.int64)
        valid = np.zeros(nw, dtype=np.int32)
        lower = np.zeros(nw, dtype=np.int32)
        any_lower = _lib.check(L.pg_pack_ascii_lower(text, len(text), codes.ctypes.data, valid.ctypes.data, lower.ctypes.data))
        return cls(torch.from_numpy(codes), torch.from_numpy(valid), len(text), off, [n for n, _ in runs],
                   valid_lower=torch.from_numpy(lower) if any_lower else None).to(device)

    # ------------------------------------------------------------------ packed-stream cache

    _MAGIC = b"PGSTRM2\0"

    def save(self, path: str) -> None:
        """write the packed stream and its runs to ``path``: a second pass over the same reads (another k, a restart, the
        other ranks of a job) then costs a file read at memcpy speed instead of a FASTQ parse.  Layout, little endian:
        magic, int64 x 8 (n_chars, n_words, n_runs, n_pairs, n_unpaired, names_bytes, mode_bytes, has_lower), mode, run_off,
        NUL-terminated names, zero padding to 4096, codes (8 B/word), valid (4 B/word), lower (4 B/word, if has_lower)."""
        codes = self.codes.cpu().numpy()
        valid = self.valid.cpu().numpy()
        names = b"".join(n.encode() + b"\0" for n in self.run_names)
        mode = self.mode.encode()
        has_lower = int(self.valid_lower is not None) | (2 if self.valid_lowq is not None else 0)      # bit 0: lower plane, bit 1: lowq plane
        head = np.array([self.n_chars, codes.size, len(self.run_names), self.n_pairs, self.n_unpaired, len(names), len(mode), has_lower], dtype="<i8")
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(self._MAGIC); f.write(head.tobytes()); f.write(mode)
            f.write(np.ascontiguousarray(self.run_off, dtype="<i8").tobytes()); f.write(names)
            f.write(b"\0" * (-f.tell() % 4096))
            codes.astype("<i8", copy=False).tofile(f)
            valid.astype("<i4", copy=False).tofile(f)
            if self.valid_lower is not None:
                self.valid_lower.cpu().numpy().astype("<i4", copy=False).tofile(f)
            if self.valid_lowq is not None:
                self.valid_lowq.cpu().numpy().astype("<i4", copy=False).tofile(f)
        os.replace(tmp, path)

    @classmethod
    def load(cls, path: str, device: str | torch.device = "cpu") -> "ReadStream":
        """a stream written by ``save`` (the arrays are memory-mapped: only what is copied to the device is read)"""
        with open(path, "rb") as f:
            if f.read(8) != cls._MAGIC:
                raise ValueError(f"{path} is not a packed read stream")
            n_chars, n_words, n_runs, n_pairs, n_unpaired, names_bytes, mode_bytes, has_lower = (int(v) for v in np.frombuffer(f.read(64), dtype="<i8"))
            mode = f.read(mode_bytes).decode()
            run_off = np.frombuffer(f.read(8 * (n_runs + 1)), dtype="<i8").astype(np.int64)
            names = f.read(names_bytes).split(b"\0")[:-1] if names_bytes else []
            at = f.tell() + (-f.tell() % 4096)
        n_extra = (has_lower & 1) + ((has_lower >> 1) & 1)
        if len(names) != n_runs or os.path.getsize(path) != at + (12 + 4 * n_extra) * n_words or n_words != words_for(n_chars):
            raise ValueError(f"{path} is truncated or inconsistent")
        codes = np.memmap(path, dtype="<i8", mode="r", offset=at, shape=(n_words,))
        valid = np.memmap(path, dtype="<i4", mode="r", offset=at + 8 * n_words, shape=(n_words,))
        lower = np.memmap(path, dtype="<i4", mode="r", offset=at + 12 * n_words, shape=(n_words,)) if has_lower & 1 else None
        lowq = np.memmap(path, dtype="<i4", mode="r", offset=at + (12 + 4 * (has_lower & 1)) * n_words, shape=(n_words,)) if has_lower & 2 else None
        device = torch.device(device)
        if device.type == "cpu":
            codes, valid = np.array(codes), np.array(valid)
            lower, lowq = (None if lower is None else np.array(lower)), (None if lowq is None else np.array(lowq))
        return cls(torch.from_numpy(codes).to(device), torch.from_numpy(valid).to(device), n_chars, run_off, [n.decode() for n in names],
                   n_pairs, n_unpaired, mode, valid_lower=None if lower is None else torch.from_numpy(lower).to(device),
                   valid_lowq=None if lowq is None else torch.from_numpy(lowq).to(device))

    def to(self, device) -> "ReadStream":
        device = torch.device(device)
        if device == self.codes.device:
            return self
        return ReadStream(self.codes.to(device), self.valid.to(device), self.n_chars, self.run_off, self.run_names,
                          self.n_pairs, self.n_unpaired, self.mode,
                          valid_lower=None if self.valid_lower is None else self.valid_lower.to(device),
                          valid_lowq=None if self.valid_lowq is None else self.valid_lowq.to(device))

    # ------------------------------------------------------------------ rows

    def rows(self, min_len: int) -> Rows:
        lens = np.diff(self.run_off)
        named = np.fromiter(map(len, self.run_names), dtype=np.int64, count=len(self.run_names)) > 0
        idx = np.nonzero(named & (lens > int(min_len)))[0].astype(np.int64)
        names = list(map(self.run_names.__getitem__, idx.tolist()))
        return Rows(idx, names, self.run_off[idx].copy(), self.run_off[idx + 1].copy())

    # ------------------------------------------------------------------ host decode (tests / FASTQ export)

    def decode(self, start: int = 0, end: int | None = None, plane: torch.Tensor | None = None) -> bytes:
        """characters [start, end) as text: bases for valid positions, 'N' for everything else (``plane``: another validity
        plane than ``valid``, e.g. ``table_valid()`` -- the reads as the multiplicity table sees them)"""
        end = self.n_chars if end is None else end
        if self.codes.is_cuda and end > start:
            # on the device, 32 M characters at a time (the tests decode whole BASELINE-size streams: numpy unpacked 0.25 GB/s)
            vp = self.valid if plane is None else plane
            sh = torch.arange(32, device=self.codes.device)
            lut = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device=self.codes.device)
            out = []
            for a in range(start, end, 1 << 25):
                b = min(end, a + (1 << 25))
                w0, w1 = a // 32, (b + 31) // 32
                code = (self.codes[w0:w1, None] >> (2 * sh)[None, :]) & 3
                ok = ((vp[w0:w1, None] >> sh[None, :]) & 1) != 0
                txt = torch.where(ok, lut[code], torch.full((), ord("N"), dtype=torch.uint8, device=self.codes.device)).reshape(-1)
                out.append(txt[a - 32 * w0:b - 32 * w0].cpu().numpy().tobytes())
            return out[0] if len(out) == 1 else b"".join(out)
        w0, w1 = start // 32, (end + 31) // 32
        c = self.codes[w0:w1].cpu().numpy().view(np.uint64)
        v = (self.valid if plane is None else plane)[w0:w1].cpu().numpy().view(np.uint32)
        sh = np.arange(32, dtype=np.uint64)
        code = ((c[:, None] >> (2 * sh)[None, :]) & np.uint64(3)).astype(np.uint8).ravel()
        ok = ((v[:, None] >> sh.astype(np.uint32)[None, :]) & np.uint32(1)).astype(bool).ravel()
        txt = np.frombuffer(b"ACTG", dtype=np.uint8)[code]
        txt = np.where(ok, txt, np.uint8(ord("N")))
        return txt[start - 32 * w0:end - 32 * w0].tobytes()
