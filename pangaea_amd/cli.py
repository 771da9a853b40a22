"""Command-line faces of the two reference counters, same flags, same output files.

    count_tnf  {-i F | -1 F -2 F} -o OUT.gz [-k 4] [-l 1000] [-t 16]              (count_tnf.cpp:117-125)
    count_kmer {-i F | -1 F -2 F} -g DUMP -o OUT.gz [-k 15] [-l 1000] [-t 16] [-v 400] [-w 10]
                                                                                  (count_kmer.cpp:112-123)
``-t`` is accepted and ignored (the GPU replaces the thread pool).  ``count_kmer -g DUMP``: an existing
jellyfish ``dump -c -t`` file is loaded with the reference's loader semantics (count_kmer.cpp:139-170); when the
file does not exist the multiplicities are counted on the GPU from the reads themselves, which is what jellyfish
would have reported.  Exit status 0, or 1 on bad arguments / failure, as the reference (cmdline.h:592-597).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np


class _Parser(argparse.ArgumentParser):
    def error(self, message):
        self.print_usage(sys.stderr)
        sys.stderr.write(f"{self.prog}: error: {message}\n")
        raise SystemExit(1)


def _common(prog: str, k_default: int) -> _Parser:
    p = _Parser(prog=prog)
    p.add_argument("-1", "--reads1", default="")
    p.add_argument("-2", "--reads2", default="")
    p.add_argument("-i", "--interleaved", default="")
    p.add_argument("-o", "--output", required=True)
    p.add_argument("-k", "--kmer", type=int, default=k_default)
    p.add_argument("-l", "--len", type=int, default=1000)
    p.add_argument("-t", "--thread", type=int, default=16)
    return p


def _inputs(a):
    if a.interleaved:
        return a.interleaved, None
    if not a.reads1 or not a.reads2:
        sys.stderr.write("Error: --reads1 and --reads2 are needed.\n")
        raise SystemExit(1)
    return a.reads1, a.reads2


def load_dump(path: str, k: int):
    """(canonical codes uint64, counts uint64) of a jellyfish text dump, later lines overriding earlier ones"""
    import pandas as pd
    df = pd.read_csv(path, sep="\t", header=None, names=["kmer", "count"], dtype={"kmer": str, "count": np.int64},
                     keep_default_na=False)
    if len(df) == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint64)
    if (df["kmer"].str.len() != k).any():
        raise ValueError(f"{path}: dump holds k-mers whose length is not {k}")
    chars = np.frombuffer("".join(df["kmer"]).encode(), dtype=np.uint8).reshape(-1, k)
    ok = np.isin(chars, np.frombuffer(b"ACGT", dtype=np.uint8)).all(axis=1)
    d = ((chars >> 1) & 3).astype(np.uint64)
    fw = np.zeros(len(df), dtype=np.uint64)
    rc = np.zeros(len(df), dtype=np.uint64)
    for j in range(k):
        fw = (fw << np.uint64(2)) | d[:, j]
        rc = rc | ((d[:, j] ^ np.uint64(2)) << np.uint64(2 * j))
    canon = np.minimum(fw, rc)[ok]
    counts = df["count"].to_numpy().astype(np.uint64)[ok]
    # last assignment wins (count_kmer.cpp:166)
    _, last = np.unique(canon[::-1], return_index=True)
    keep = len(canon) - 1 - last
    return canon[keep], counts[keep]


def main_count_tnf(argv=None) -> int:
    a = _common("count_tnf", 4).parse_args(argv)
    r1, r2 = _inputs(a)
    from . import feature
    try:
        names, tnf, _ = feature.compute_features(r1, r2, 0, a.kmer, 1, 1, a.len, want_abd=False)
        feature.write_csv_gz(a.output, names, tnf)
    except Exception as e:
        sys.stderr.write(f"count_tnf: {e}\n")
        return 1
    return 0


def main_count_kmer(argv=None) -> int:
    p = _common("count_kmer", 15)
    p.add_argument("-g", "--global", dest="global_", required=True)
    p.add_argument("-v", "--vector", type=int, default=400)
    p.add_argument("-w", "--window", type=int, default=10)
    a = p.parse_args(argv)
    print(a.interleaved)                       # the reference echoes the interleaved path first (count_kmer.cpp:173)
    r1, r2 = _inputs(a)
    import torch
    from . import feature
    from .kmer import KmerTable
    try:
        table = None
        if os.path.isfile(a.global_):
            codes, counts = load_dump(a.global_, a.kmer)
            table = KmerTable.from_items(a.kmer, codes, counts, torch.device("cuda", torch.cuda.current_device()))
        names, _, abd = feature.compute_features(r1, r2, a.kmer, 0, a.window, a.vector, a.len, want_tnf=False, table=table)
        feature.write_csv_gz(a.output, names, abd)
    except Exception as e:
        sys.stderr.write(f"count_kmer: {e}\n")
        return 1
    return 0
