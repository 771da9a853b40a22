"""``Feature`` -- drop-in for /root/reference/src/feature.py:11-146 on the GPU.

Same constructor ``Feature(args, script_path)``, same methods (``extract_features``, ``load_features``,
``run_jellyfish``, ``calcu_tnf``), same return triple ``(names, abundance[N,V], tnf[N,D])`` and the same cache files
under ``<output>/1.features``:
    tnf.m{minl}.gz / .pkl                         (feature.py:126-127)
    abundance.k{k}.v{V}.w{W}.m{minl}.gz / .pkl    (feature.py:68-69)
    feature_finished                              (feature.py:37-38)
The reference obtains the two matrices from three subprocesses (jellyfish, count_kmer, count_tnf) running in two
threads; here one ingest + one table build + one fused feature launch produce both, and the two ``calcu_*``
methods only differ in which cache file they write.  ``abundance.k{k}.count/.dump`` (jellyfish's own files) are
not produced: the multiplicity table lives in HBM.

Values are returned exactly as the reference returns them, i.e. as pandas would re-read the CSV cache
(feature.py:113-123): counts >= 1 000 000 pass through ``%g`` (six significant digits) and turn their column into
float64, and the name column goes through pandas' type inference.
"""
from __future__ import annotations

import io
import logging
import os

import numpy as np
import pandas as pd
import torch

from . import _lib, dist as pdist, runtime as _runtime
from .kmer import KmerTable, Plan, count_kmers, features, tnf_ncols
from .reads import ReadStream


def g_roundtrip(mat: np.ndarray) -> np.ndarray:
    """what ``ostream << double`` followed by ``pd.read_csv`` leaves of an integer matrix: unchanged int64 unless a
    count needs more than six significant digits in %g form (>= 1e6), in which case those entries are rounded and
    the matrix becomes float64 (pandas: any float column makes ``to_numpy()`` float)"""
    mat = np.asarray(mat, dtype=np.int64)
    big = mat >= 1_000_000
    if not big.any():
        return mat
    out = mat.astype(np.float64)
    out[big] = [float("%g" % v) for v in mat[big]]
    return out


def frame_like_read_csv(names, mat: np.ndarray) -> pd.DataFrame:
    """the DataFrame ``pd.read_csv(<cache>.gz, header=None)`` would build from the rows we hold in memory"""
    if len(names):
        col0 = pd.read_csv(io.StringIO("\n".join(names) + "\n"), header=None, skip_blank_lines=False)[0]
    else:
        col0 = pd.Series([], dtype=object)
    rt = g_roundtrip(mat)
    cols = {0: col0.to_numpy()}
    if rt.dtype == np.float64:
        imat = np.asarray(mat, dtype=np.int64)
        for j in range(rt.shape[1]):
            cols[j + 1] = rt[:, j] if (imat[:, j] >= 1_000_000).any() else imat[:, j]
    else:
        for j in range(rt.shape[1]):
            cols[j + 1] = rt[:, j]
    return pd.DataFrame(cols)


def write_csv_gz(path: str, names, mat: np.ndarray) -> None:
    """the cache file, complete or absent: written next to its place and renamed (a reader -- or a later resume -- never sees
    a half-written file)"""
    mat = np.ascontiguousarray(mat, dtype=np.int32)
    blob = b"".join(str(n).encode() + b"\0" for n in names)
    tmp = f"{path}.tmp{os.getpid()}"
    try:
        _lib.check(_lib.load().pg_write_csv_gz(tmp.encode(), blob, mat.ctypes.data, mat.shape[0], mat.shape[1] if mat.ndim == 2 else 0))
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def to_pickle_atomic(frame: pd.DataFrame, path: str) -> None:
    tmp = f"{path}.tmp{os.getpid()}"
    try:
        frame.to_pickle(tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def _ingest(reads1: str, reads2: str | None, world: int, stream_cache: str | None, device="cpu") -> ReadStream:
    """this rank's stream (on ``device`` already when the ingest could copy it there piece by piece, `pg_ingest_fastq_device`): from the packed-stream cache when one is given and is newer than the reads, else from the
    FASTQ file(s) (and the cache is written for the next pass)"""
    cache = None
    if stream_cache:
        rank = torch.distributed.get_rank() if world > 1 else 0
        cache = f"{stream_cache}.r{rank}of{world}.pgstream"
        newest = max(os.path.getmtime(p) for p in (reads1, reads2) if p)
        fresh = os.path.exists(cache) and os.path.getmtime(cache) >= newest
        if world > 1:
            fresh = pdist.everyone(fresh)        # the sharded ingest is a collective: all ranks take the same branch
        if fresh:
            logging.info(f"packed read stream from {cache}")
            return ReadStream.load(cache)
    part = pdist.ingest_shard(reads1, reads2, device=device) if world > 1 else ReadStream.from_fastq(reads1, reads2, device=device)
    if cache:
        part.save(cache)
    return part


def _sharded_mini_applies(stream, plan, k, window, vsize, lowercase_is_base) -> bool:
    """may the multi-rank super-k-mer form (``dist.MiniSharded``) take this input?  Packed slots (13 <= k <= 21), rows the
    partition records can name, exact bins, no soft-masked / quality-masked planes -- and the SAME answer on every rank."""
    from . import _lib
    ok = (_lib.MINI_MIN_K <= k <= _lib.HASH_MAX_K and plan.shuffle_ok and 0 < plan.n_rows < (1 << 17)
          and 1 <= vsize <= _lib.SHUFFLE_MAX_VSIZE and window >= 1 and window * vsize <= _lib.HASH_COUNT_SAT
          and stream.table_valid(lowercase_is_base) is stream.valid and stream.rows_inside_table
          and os.environ.get("PANGAEA_NO_MINI", "0") in ("", "0"))
    return pdist.everyone(ok)


def compute_features(reads1: str, reads2: str | None, k: int, k_tnf: int, window: int, vsize: int, min_len: int,
                     device=None, want_tnf: bool = True, want_abd: bool = True, table: KmerTable | None = None,
                     stream_cache: str | None = None, lowercase_is_base: bool = True, gather: str = "all"):
    """(names, tnf int32 ndarray or None, abd int32 ndarray or None) of a barcode-sorted FASTQ, on the GPU.

    Under an initialised ``torch.distributed`` group every rank takes a contiguous range of runs and the table is
    exchanged once; the ROWS stay where they were made.  ``gather``:
      "none"   every rank returns its own block of rows (rank order = file order);
      "rank0"  rank 0 additionally receives all blocks (it writes the cache files) and returns
               (names, tnf, abd, local) with ``local`` = its own block; the other ranks return their block;
      "all"    every rank returns the full matrices (single-call convenience: the CLI tools)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    world = torch.distributed.get_world_size() if pdist.is_distributed() else 1
    from . import kmer as _kmer
    # the scratch of the counting pipeline is allocated while the host threads parse (hipMalloc of tens of GB is most of what a
    # FIRST pass over a data set costs besides its kernels): sizes from the file size, ~690 bytes of plain FASTQ per 150 bp pair
    warm = None
    if want_abd and table is None and not any(str(p).endswith(".gz") for p in (reads1, reads2) if p):
        try:
            fbytes = sum(os.path.getsize(p) for p in (reads1, reads2) if p)
            warm = _kmer.prewarm_workspaces(device, int(fbytes / 690 / world * 1.03) + 1, k, vsize)
        except OSError:
            warm = None
    stream = _ingest(reads1, reads2, world, stream_cache, device).to(device)
    if warm is not None:
        warm.join()
    # one rank: the table's sizing pass (a HyperLogLog sketch of the distinct k-mers) is launched first and runs on the GPU while
    # the host assembles the rows
    sketch = None
    if world == 1 and want_abd and table is None and _kmer.KmerTable.default_kind(k) != "dense":
        sketch = _kmer.distinct_sketch(stream, k, lowercase_is_base=lowercase_is_base)
    rows = stream.rows(min_len)
    plan = Plan(rows, device)
    tnf = abd = None
    if want_abd and table is None and world > 1 and _sharded_mini_applies(stream, plan, k, window, vsize, lowercase_is_base):
        # several ranks: the super-k-mer pipeline on every rank's own reads, entries to bucket owners, bins back
        tnf, abd, _ = pdist.features_sharded_mini(stream, plan, k, k_tnf if want_tnf else None, window, vsize)
    else:
        if want_abd and table is None:
            table = (pdist.count_kmers_sharded(stream, k, rows=plan, lowercase_is_base=lowercase_is_base) if world > 1
                     else count_kmers(stream, k, rows=plan, emit=(window, vsize), lowercase_is_base=lowercase_is_base,
                                      distinct_hint=None if sketch is None else max(1 << 13, int(1.1 * _kmer.sketch_estimate(sketch))),
                                      load=None if sketch is None else 0.4))
        tnf, abd = features(stream, plan, k_tnf=k_tnf if want_tnf else None, table=table if want_abd else None,
                            window=window, vsize=vsize)
    names = list(rows.names)
    host = lambda t: t.cpu().numpy() if t is not None else None
    if world == 1 or gather == "none":
        return names, host(tnf), host(abd)
    local = (names, host(tnf), host(abd))
    rank = torch.distributed.get_rank()
    if gather == "all":
        all_names = [None] * world
        torch.distributed.all_gather_object(all_names, names)
        full_names = [n for part in all_names for n in part]
        sizes = [len(part) for part in all_names]
        full = []
        for m in (tnf, abd):
            if m is None:
                full.append(None)
                continue
            got = None
            for dst in range(world):                     # (rare path: one gather per destination, rows stay int32)
                part = pdist.gather_rows(m, dst=dst)
                got = part if dst == rank else got
            assert got.shape[0] == sum(sizes)
            full.append(host(got))
        return full_names, full[0], full[1]
    gathered_names = [None] * world if rank == 0 else None
    torch.distributed.gather_object(names, gathered_names, dst=0)
    g_tnf = pdist.gather_rows(tnf, dst=0) if tnf is not None else None
    g_abd = pdist.gather_rows(abd, dst=0) if abd is not None else None
    if rank != 0:
        return local
    return [n for part in gathered_names for n in part], host(g_tnf), host(g_abd), local


class Feature:
    def __init__(self, args, script_path):
        self.args = args
        self.tnf_k = str(args.tnf_kmer)
        self.ws = args.window_size
        self.vs = args.vector_size
        self.kmer = args.kmer
        self.minl = args.min_length
        self.threads = args.threads
        self.script_path = script_path
        self.feature_dir = os.path.join(args.output, "1.features")
        os.makedirs(self.feature_dir, exist_ok=True)
        self._cache = None          # (names, tnf int32, abd int32) of the fused GPU pass
        # several ranks (torchrun): every rank computes the rows of its own runs; rank 0 alone receives all rows and writes the
        # cache files and the marker; ``local`` keeps this rank's block (names, tnf, abd) for the replicated encode of step 2
        self.world = torch.distributed.get_world_size() if pdist.is_distributed() else 1
        self.rank = torch.distributed.get_rank() if self.world > 1 else 0
        self.local = None

    # ------------------------------------------------------------------ paths (same names as the reference)

    def _abd_paths(self):
        stem = os.path.join(self.feature_dir, f"abundance.k{self.kmer}.v{self.vs}.w{self.ws}.m{self.minl}")
        return stem + ".gz", stem + ".pkl"

    def _tnf_paths(self):
        stem = os.path.join(self.feature_dir, f"tnf.m{self.minl}")
        return stem + ".gz", stem + ".pkl"

    def _inputs(self):
        if self.args.reads1 and self.args.reads2:
            return self.args.reads1, self.args.reads2
        if self.args.interleaved_reads:
            return self.args.interleaved_reads, None
        raise ValueError("reads must be specified")

    def _compute(self, want_tnf=True, want_abd=True):
        if self._cache is None:
            r1, r2 = self._inputs()
            logging.info("GPU feature pass: ingest + k-mer table + TNF/abundance rows")
            # PANGAEA_STREAM_CACHE=1 keeps the packed read stream next to the feature caches (1.features/reads.*.pgstream)
            cache = os.path.join(self.feature_dir, "reads") if os.environ.get("PANGAEA_STREAM_CACHE", "0") not in ("", "0") else None
            out = compute_features(r1, r2, int(self.kmer), int(self.tnf_k), int(self.ws), int(self.vs), int(self.minl),
                                           device=getattr(self.args, "device", None), want_tnf=want_tnf, want_abd=want_abd,
                                           stream_cache=cache, gather="rank0",
                                           # jellyfish's rules for the multiplicity table (feature.py:76-94): lower-case bases
                                           # count (soft-masked input; PANGAEA_LOWERCASE_IS_BASE=0 turns that off), bases
                                           # below --min-qual-char=? of paired files do not (ReadStream.table_valid)
                                           lowercase_is_base=os.environ.get("PANGAEA_LOWERCASE_IS_BASE", "1") not in ("", "0"))
            if self.world > 1:
                self.local = out[3] if self.rank == 0 else out
                out = out[:3]
            self._cache = out
        return self._cache

    # ------------------------------------------------------------------ the reference's public methods

    def extract_features(self):
        # (the network comes next, pangaea.py:70,90: its GEMM library initialises under the ingest -- see runtime.warm_blas; the
        # count_tnf / count_kmer tools, which end with the matrices, do not ask for it)
        if torch.cuda.is_available():
            _runtime.warm_blas(getattr(self.args, "device", None) or torch.device("cuda", torch.cuda.current_device()))
        if self.world > 1:
            return self._extract_features_sharded()
        readnames1, abundance = self.run_jellyfish()
        readnames2, tnf = self.calcu_tnf()
        assert (readnames1 == readnames2).all()
        with open(os.path.join(self.feature_dir, "feature_finished"), "w") as f:
            f.write("feature finished")
        return readnames1, abundance, tnf

    def _extract_features_sharded(self):
        """several ranks: the GPU pass is collective (sharded ingest, one table exchange, rows gathered to rank 0); the files
        are rank 0's business alone -- the others wait for the marker's barrier and return their own block of rows in the
        reference's types (what pandas would have read back)"""
        abd_gz, abd_pkl = self._abd_paths()
        tnf_gz, tnf_pkl = self._tnf_paths()
        have = all(os.path.isfile(p) for p in (abd_gz, abd_pkl, tnf_gz, tnf_pkl))
        if not pdist.everyone(have):                 # (a collective decision: no rank may skip the pass on its own)
            self._compute()
        result = None
        if self.rank == 0:
            readnames1, abundance = self.run_jellyfish()
            readnames2, tnf = self.calcu_tnf()
            assert (readnames1 == readnames2).all()
            with open(os.path.join(self.feature_dir, "feature_finished"), "w") as f:
                f.write("feature finished")
            result = (readnames1, abundance, tnf)
        pdist.wait_for_all()                         # the caches and the marker are complete before anybody moves on (control plane:
                                                     # writing them can take rank 0 longer than a data collective may wait)
        if self.rank != 0:
            if self.local is None:
                return np.zeros(0, dtype=object), np.zeros((0, int(self.vs)), dtype=np.int64), np.zeros((0, 0), dtype=np.int64)
            names, tnf, abd = self.local
            fa, ft = frame_like_read_csv(names, abd), frame_like_read_csv(names, tnf)
            result = (fa[0].to_numpy(), fa.drop(columns=0).to_numpy(), ft.drop(columns=0).to_numpy())
        return result

    def _materialise(self, gz: str, pkl: str, which: int, what: str):
        """the reference's per-artifact resume logic (feature.py:104-123, 129-146): write <cache>.gz unless it is
        there, write <cache>.pkl unless it is there, and return what the pickle holds"""
        fresh = None
        if not os.path.isfile(gz):
            logging.info(f"caculate {what} : {gz}")
            names, tnf, abd = self._compute()
            mat = (tnf, abd)[which]
            write_csv_gz(gz, names, mat)
            fresh = frame_like_read_csv(names, mat)
        if not os.path.isfile(pkl):
            frame = fresh if fresh is not None else pd.read_csv(gz, header=None)
            to_pickle_atomic(frame, pkl)
        else:
            logging.info(f"load {what}")
            frame = pd.read_pickle(pkl)
        names = frame[0].to_numpy()
        mat = frame.drop(columns=0).to_numpy()
        logging.info(f"{what} shape {mat.shape}")
        return names, mat

    def run_jellyfish(self):
        gz, pkl = self._abd_paths()
        return self._materialise(gz, pkl, 1, "abundance")

    def calcu_tnf(self):
        gz, pkl = self._tnf_paths()
        return self._materialise(gz, pkl, 0, "tnf")

    def load_features(self):
        abd_gz, abd_pkl = self._abd_paths()
        _, tnf_pkl = self._tnf_paths()
        try:
            tnf = pd.read_pickle(tnf_pkl)
            readnames = tnf[0].to_numpy()
            tnf = tnf.drop(columns=0).to_numpy()
            logging.info(f"tnf shape {tnf.shape}")
        except Exception:
            raise Exception(tnf_pkl, " file not found")
        if os.path.isfile(abd_pkl):
            logging.info("load features from pickle file " + abd_pkl)
            df = pd.read_pickle(abd_pkl)
        elif os.path.isfile(abd_gz):
            # the reference re-reads the comma file with sep="\t" here (feature.py:58-61) and cannot succeed;
            # the .pkl is always written next to the .gz, so this branch only serves foreign caches
            logging.info(abd_pkl + " not found, load features from " + abd_gz)
            df = pd.read_csv(abd_gz, header=None)
        else:
            raise Exception(abd_gz, " file not found")
        readnames = df[0].to_numpy()
        abundance = df.drop(columns=0).to_numpy()
        logging.info(f"abundance shape {abundance.shape}")
        return readnames, abundance, tnf
