"""Bin writer: ``clusters.tsv`` -> ``cluster_bin<label>.fq`` / ``.barcode`` (the reference's ``extract_reads`` tool,
src/cpptools/extract_reads.cpp; called from clustering.py:119-122).  Host I/O only -- implemented in the C-ABI library."""
from __future__ import annotations

import ctypes as C

from . import _lib


def extract_reads(reads1: str, reads2: str | None, clusters_tsv: str, out_prefix: str) -> int:
    """returns the number of read pairs written"""
    n = C.c_int64(0)
    _lib.check(_lib.load().pg_extract_reads(str(reads1).encode(), str(reads2).encode() if reads2 else None,
                                            str(clusters_tsv).encode(), str(out_prefix).encode(), C.byref(n)))
    return int(n.value)
