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

from . import _lib, dist as pdist
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
    mat = np.ascontiguousarray(mat, dtype=np.int32)
    blob = b"".join(str(n).encode() + b"\0" for n in names)
    _lib.check(_lib.load().pg_write_csv_gz(path.encode(), blob, mat.ctypes.data, mat.shape[0], mat.shape[1] if mat.ndim == 2 else 0))


def _ingest(reads1: str, reads2: str | None, world: int, stream_cache: str | None) -> ReadStream:
    """this rank's host stream: from the packed-stream cache when one is given and is newer than the reads, else from the
    FASTQ file(s) (and the cache is written for the next pass)"""
    cache = None
    if stream_cache:
        rank = torch.distributed.get_rank() if world > 1 else 0
        cache = f"{stream_cache}.r{rank}of{world}.pgstream"
        newest = max(os.path.getmtime(p) for p in (reads1, reads2) if p)
        if os.path.exists(cache) and os.path.getmtime(cache) >= newest:
            logging.info(f"packed read stream from {cache}")
            return ReadStream.load(cache)
    part = pdist.ingest_shard(reads1, reads2) if world > 1 else ReadStream.from_fastq(reads1, reads2)
    if cache:
        part.save(cache)
    return part


def compute_features(reads1: str, reads2: str | None, k: int, k_tnf: int, window: int, vsize: int, min_len: int,
                     device=None, want_tnf: bool = True, want_abd: bool = True, table: KmerTable | None = None,
                     stream_cache: str | None = None, lowercase_is_base: bool = True):
    """(names, tnf int32 ndarray or None, abd int32 ndarray or None) of a barcode-sorted FASTQ, on the GPU.
    Under an initialised ``torch.distributed`` group every rank takes a contiguous range of runs, the table is
    exchanged once, and the rows are gathered so every rank returns the full matrices."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    world = torch.distributed.get_world_size() if pdist.is_distributed() else 1
    stream = _ingest(reads1, reads2, world, stream_cache).to(device)
    rows = stream.rows(min_len)
    plan = Plan(rows, device)
    if want_abd and table is None:
        table = (pdist.count_kmers_sharded(stream, k, rows=plan, lowercase_is_base=lowercase_is_base) if world > 1
                 else count_kmers(stream, k, rows=plan, emit=(window, vsize), lowercase_is_base=lowercase_is_base))
    tnf, abd = features(stream, plan, k_tnf=k_tnf if want_tnf else None, table=table if want_abd else None,
                        window=window, vsize=vsize)
    names = list(rows.names)
    if world > 1:
        all_names = [None] * world
        torch.distributed.all_gather_object(all_names, names)
        names = [n for part_names in all_names for n in part_names]
        if tnf is not None:
            tnf = torch.cat([p.view(-1, tnf.shape[1]) for p in pdist.gather_pairs(tnf.reshape(-1).to(torch.int64))]).to(torch.int32)
        if abd is not None:
            abd = torch.cat([p.view(-1, abd.shape[1]) for p in pdist.gather_pairs(abd.reshape(-1).to(torch.int64))]).to(torch.int32)
    return names, (tnf.cpu().numpy() if tnf is not None else None), (abd.cpu().numpy() if abd is not None else None)


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
            self._cache = compute_features(r1, r2, int(self.kmer), int(self.tnf_k), int(self.ws), int(self.vs), int(self.minl),
                                           device=getattr(self.args, "device", None), want_tnf=want_tnf, want_abd=want_abd,
                                           stream_cache=cache,
                                           # jellyfish's rules for the multiplicity table (feature.py:76-94): lower-case bases
                                           # count (soft-masked input; PANGAEA_LOWERCASE_IS_BASE=0 turns that off), bases
                                           # below --min-qual-char=? of paired files do not (ReadStream.table_valid)
                                           lowercase_is_base=os.environ.get("PANGAEA_LOWERCASE_IS_BASE", "1") not in ("", "0"))
        return self._cache

    # ------------------------------------------------------------------ the reference's public methods

    def extract_features(self):
        readnames1, abundance = self.run_jellyfish()
        readnames2, tnf = self.calcu_tnf()
        assert (readnames1 == readnames2).all()
        with open(os.path.join(self.feature_dir, "feature_finished"), "w") as f:
            f.write("feature finished")
        return readnames1, abundance, tnf

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
            frame.to_pickle(pkl)
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
