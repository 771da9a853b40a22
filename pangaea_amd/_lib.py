"""ctypes binding of libpangaea_feat.so (include/pangaea_feat.h).

The HIP library is the product path: there is no CPU fallback.  ``load()`` raises when the shared
object is missing (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C pangaea_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Which library: libpangaea_feat.so, the product, unless PANGAEA_LIB says otherwise --
#   PANGAEA_LIB=checked   the build whose super-k-mer kernels check every global store against the capacity of its buffer
#                         (make -C pangaea_amd/csrc checked; include/pangaea_feat.h: PG_STATUS_BOUNDS): slower, tests of new kernel code
#   PANGAEA_LIB=<path>    a variant (make -C pangaea_amd/csrc variant NAME=.. KFLAGS=..): timing experiments, phase stamps.  A variant
#                         may compute wrong results, so load() refuses one (pg_build_flags() != 0) without PANGAEA_ALLOW_VARIANT=1.
# Nothing ever has to be copied over the product library to try another build.
_WANT = os.environ.get("PANGAEA_LIB", "")
LIB_PATH = (os.path.join(_HERE, "libpangaea_feat.so") if not _WANT else os.path.join(_HERE, "libpangaea_feat_checked.so") if _WANT == "checked"
            else os.path.abspath(_WANT))
BUILD_CHECKED, BUILD_STAMPS, BUILD_VARIANT = 1, 2, 4
STATUS_TABLE_FULL, STATUS_OVERFLOW_LIST, STATUS_PLAN_MISMATCH, STATUS_BOUNDS = 1, 2, 4, 8
_LIB = None

PG_OK = 0
PG_ETABLEFULL = -6
TABLE_DENSE, TABLE_HASH, TABLE_WIDE, TABLE_MINI, TABLE_MINI_WIDE = 1, 2, 3, 4, 5
DENSE_MAX_K, HASH_MAX_K, WIDE_MAX_K = 16, 21, 31
HASH_COUNT_BITS = 22
HASH_COUNT_SAT = 1 << 21
TNF_MAX_K = 6
WORD_ALIGN = 256
BUCKET_MAX_LOG2_SLOTS, BUCKET_MAX_LOG2_BUCKETS = 14, 17
ABI_VERSION = 9
MINI_MIN_K, MINI_MAX_LOG2_BUCKETS, MINI_MAX_ROWS, MINI_WIDE_MAX_LOG2_BUCKET_SLOTS = 13, 16, (1 << 20) - 2, 13
SHUFFLE_MAX_VSIZE = 512
DEFERRED_MAX_GROUP_LOG2 = 3
KEY42_M1 = 0x3d7ed558ccd          # pg_key42 (include/pangaea_feat.h)
KEY42_M2 = 0x1fe1a85ec53
HLL_REGISTERS = 4096
MAX_ROWS = (1 << 22) - 2


class PangaeaError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpangaea_feat: {msg} (status {code})")
        self.code = code


class pg_table(C.Structure):
    _fields_ = [("kind", C.c_int32), ("k", C.c_int32), ("log2_slots", C.c_int32), ("log2_bucket_slots", C.c_int32),
                ("data", C.c_void_p)]


class pg_rows(C.Structure):
    _fields_ = [("row_start", C.c_void_p), ("row_end", C.c_void_p), ("n_rows", C.c_int64), ("strict_valid", C.c_void_p)]


def build() -> None:
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc")], check=True)


def load() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (make -C pangaea_amd/csrc). "
            "There is no CPU fallback for the feature path.")
    # torch first: its wheel carries its own libamdhip64, and the library's kernels run on torch's device memory and streams --
    # both must sit on ONE HIP runtime.  Loaded before torch, this library binds /opt/rocm's copy, torch then brings a second
    # runtime into the process and the library's launches fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, cp = C.c_void_p, C.c_int64, C.c_int, C.c_char_p
    tp = C.POINTER(pg_table)
    rp = C.POINTER(pg_rows)
    sig = {
        "pg_abi_version": (i32, []),
        "pg_build_flags": (C.c_uint32, []),
        "pg_last_error": (cp, []),
        "pg_device_count": (i32, []),
        "pg_ingest_fastq": (i32, [cp, cp, C.POINTER(vp)]),
        "pg_fastq_count_newlines": (i32, [cp, i32, i32, C.POINTER(i64)]),
        "pg_ingest_fastq_shard": (i32, [cp, i32, i32, C.POINTER(i64), C.POINTER(vp)]),
        "pg_ingest_staging_words": (i64, [i64]),
        "pg_inflate_to_memfd": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(i64)]),
        "pg_ingest_fastq_device": (i32, [cp, i32, i32, C.POINTER(i64), i64, vp, vp, i64, C.POINTER(vp)]),
        "pg_ingest_place": (i32, [vp, vp, vp, i64, vp, vp, i64, vp]),
        "pg_ingest_pair_staging_words": (i64, [i64, i64]),
        "pg_ingest_fastq_pair_device": (i32, [C.c_char_p, C.c_char_p, i64, i64, vp, vp, vp, i64, C.POINTER(vp)]),
        "pg_reads_staged_lowq": (i32, [vp]),
        "pg_ingest_place_pair": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, i64, vp]),
        "pg_set_ingest_threads": (None, [i32]),
        "pg_reads_free": (None, [vp]),
        "pg_reads_n_chars": (i64, [vp]),
        "pg_reads_n_words": (i64, [vp]),
        "pg_reads_n_pairs": (i64, [vp]),
        "pg_reads_n_unpaired": (i64, [vp]),
        "pg_reads_n_runs": (i64, [vp]),
        "pg_reads_codes": (vp, [vp]),
        "pg_reads_valid": (vp, [vp]),
        "pg_reads_lower": (vp, [vp]),
        "pg_reads_lowq": (vp, [vp]),
        "pg_reads_run_off": (vp, [vp]),
        "pg_reads_run_name": (cp, [vp, i64]),
        "pg_reads_run_names": (i64, [vp, C.c_char_p, i64]),
        "pg_reads_mode": (cp, [vp]),
        "pg_reads_rows": (i64, [vp, i32, vp]),
        "pg_words_for": (i64, [i64]),
        "pg_pack_ascii": (i32, [cp, i64, vp, vp]),
        "pg_pack_ascii_lower": (i32, [cp, i64, vp, vp, vp]),
        "pg_plan_segments": (i64, [vp, vp, i64, i64, vp, vp, vp]),
        "pg_tnf_ncols": (i32, [i32]),
        "pg_tnf_colmap": (i32, [i32, vp, vp]),
        "pg_kmer_count": (i32, [vp, vp, i64, i64, tp, vp, vp]),
        "pg_kmer_distinct_sketch": (i32, [vp, vp, i64, i64, i32, vp, vp]),
        "pg_kmer_count_workspace_bytes": (i64, [i64, tp]),
        "pg_kmer_count_bucketed": (i32, [vp, vp, i64, i64, tp, i32, rp, vp, i64, vp, vp]),
        "pg_abundance_workspace_bytes": (i64, [i64, i64, i32, tp]),
        "pg_abundance_from_records": (i32, [tp, rp, i32, i32, vp, vp, i64, i64, vp, i64, vp]),
        "pg_abundance_from_emitted": (i32, [tp, rp, i32, i32, vp, vp, i64, i64, vp, i64, vp]),
        "pg_kmer_count_bucketed_emit": (i32, [vp, vp, i64, i64, tp, rp, vp, i64, i32, i32, vp, i64, vp, vp]),
        "pg_kmer_merge": (i32, [vp, i64, tp, vp, vp]),
        "pg_kmer_merge_bucketed": (i32, [vp, vp, i32, tp, vp, vp]),
        "pg_kmer_rebuild_bucketed": (i32, [vp, vp, i32, tp, vp, vp]),
        "pg_kmer_rebuild_bucketed_range": (i32, [vp, vp, i32, tp, i64, i64, vp, vp]),
        "pg_table_bucket_fill": (i32, [tp, vp, vp]),
        "pg_kmer_count_deferred": (i32, [vp, vp, i64, i64, tp, i32, rp, vp, i64, vp, vp, vp]),
        "pg_deferred_gather": (i32, [tp, vp, i64, i64, vp, vp, vp, vp]),
        "pg_deferred_gather_planes": (i32, [tp, vp, i64, i64, vp, vp, vp, vp, vp, vp, i64, vp, vp]),
        "pg_kmer_rebuild_planes_range": (i32, [vp, i64, i64, vp, i32, tp, i64, i64, vp, vp]),
        "pg_table_bucket_fill_range": (i32, [tp, i64, i64, vp, vp]),
        "pg_table_compact_planes_range": (i32, [tp, i64, i64, vp, vp, vp, vp, vp, i64, vp, vp]),
        "pg_table_compact": (i32, [tp, vp, vp, vp]),
        "pg_kmer_merge_wide": (i32, [vp, vp, i64, tp, vp, vp]),
        "pg_mini_plan_bytes": (i64, [i64, tp]),
        "pg_mini_plan": (i32, [vp, vp, i64, i64, tp, rp, vp, i64, vp]),
        "pg_mini_records_bytes": (i64, [i64, tp]),
        "pg_mini_shuffle_bytes": (i64, [i64, i64, i32]),
        "pg_mini_shuffle_bytes_merged": (i64, [i64, i64, i32, tp]),
        "pg_mini_count": (i32, [vp, vp, i64, i64, tp, rp, vp, i64, vp, i64, i32, i32, vp, i64, vp, i64, vp, vp]),
        "pg_mini_merge_words": (i64, [i64, i64, i64, tp]),
        "pg_mini_count_piece": (i32, [vp, vp, i64, i64, tp, rp, vp, i64, vp, i64, i32, i32, vp, i64, i32, vp, vp]),
        "pg_mini_lookup_begin": (i32, [tp, rp, i64, i32, vp, i64, vp]),
        "pg_mini_lookup_piece": (i32, [tp, rp, vp, i64, i64, vp, i64, i32, i32, vp, i64, vp, vp, vp]),
        "pg_mini_half_bytes": (i64, [tp]),
        "pg_mini_count_half": (i32, [vp, vp, i64, i64, tp, rp, vp, i64, vp, i64, i32, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp]),
        "pg_mini_gather_entries": (i32, [tp, vp, i64, vp, vp, vp, i64, vp, vp]),
        "pg_mini_merge_bins": (i32, [vp, i64, vp, i32, tp, i64, i64, i32, i32, vp, vp, vp]),
        "pg_mini_lookup_half": (i32, [tp, rp, vp, i64, vp, i64, i64, i32, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp]),
        "pg_mini_abundance_from_emitted": (i32, [tp, rp, i32, vp, vp, i64, i64, vp, i64, vp]),
        "pg_mini_wait_first_pass": (i32, [vp]),
        "pg_features": (i32, [vp, vp, i64, vp, vp, vp, i64, i32, vp, vp, tp, i32, i32, vp, vp]),
        "pg_normalize_rows": (i32, [vp, i64, i32, vp, vp, vp]),
        "pg_write_csv_gz": (i32, [cp, cp, vp, i64, i64]),
        "pg_extract_reads": (i32, [cp, cp, cp, cp, C.POINTER(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)           # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    if L.pg_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {L.pg_abi_version()} != {ABI_VERSION}")
    # the product is flags == 0; a checked library only when it was asked for by name; anything else (a timing experiment may
    # compute wrong results) only with PANGAEA_ALLOW_VARIANT=1 -- whatever the file is called
    flags = int(L.pg_build_flags())
    allowed = BUILD_CHECKED if _WANT == "checked" else 0
    if flags & ~allowed and os.environ.get("PANGAEA_ALLOW_VARIANT", "") != "1":
        raise RuntimeError(f"{LIB_PATH}: pg_build_flags() = {flags} (1 checked, 2 stamps, 4 variant) -- not the product build; "
                           "rebuild it (make -C pangaea_amd/csrc) or set PANGAEA_ALLOW_VARIANT=1 for an experiment")
    _LIB = L
    return L


EXPORTS = ["pg_abi_version", "pg_build_flags", "pg_last_error", "pg_device_count", "pg_ingest_fastq", "pg_fastq_count_newlines", "pg_ingest_fastq_shard", "pg_ingest_staging_words", "pg_inflate_to_memfd", "pg_ingest_fastq_device", "pg_ingest_place", "pg_ingest_pair_staging_words", "pg_ingest_fastq_pair_device", "pg_reads_staged_lowq", "pg_ingest_place_pair",
           "pg_set_ingest_threads", "pg_reads_free", "pg_reads_n_chars",
           "pg_reads_n_words", "pg_reads_n_pairs", "pg_reads_n_unpaired", "pg_reads_n_runs", "pg_reads_codes",
           "pg_reads_valid", "pg_reads_lower", "pg_reads_lowq", "pg_reads_run_off", "pg_reads_run_name", "pg_reads_run_names", "pg_reads_mode", "pg_reads_rows", "pg_words_for",
           "pg_pack_ascii", "pg_pack_ascii_lower", "pg_plan_segments", "pg_tnf_ncols", "pg_tnf_colmap", "pg_kmer_count", "pg_kmer_distinct_sketch", "pg_kmer_count_workspace_bytes",
           "pg_kmer_count_bucketed", "pg_kmer_merge", "pg_kmer_merge_bucketed", "pg_kmer_rebuild_bucketed", "pg_kmer_rebuild_bucketed_range", "pg_table_bucket_fill", "pg_kmer_count_deferred", "pg_deferred_gather", "pg_deferred_gather_planes", "pg_kmer_rebuild_planes_range", "pg_table_bucket_fill_range", "pg_table_compact_planes_range", "pg_table_compact", "pg_kmer_merge_wide", "pg_abundance_workspace_bytes",
           "pg_abundance_from_records", "pg_abundance_from_emitted", "pg_kmer_count_bucketed_emit",
           "pg_mini_plan_bytes", "pg_mini_plan", "pg_mini_records_bytes", "pg_mini_shuffle_bytes", "pg_mini_shuffle_bytes_merged", "pg_mini_merge_words", "pg_mini_count_piece", "pg_mini_lookup_begin", "pg_mini_lookup_piece", "pg_mini_count", "pg_mini_half_bytes", "pg_mini_count_half", "pg_mini_gather_entries", "pg_mini_merge_bins", "pg_mini_lookup_half", "pg_mini_wait_first_pass", "pg_mini_abundance_from_emitted",
           "pg_features", "pg_normalize_rows", "pg_write_csv_gz", "pg_extract_reads"]


def check(rc: int) -> int:
    """raise on a negative status; pass through counts"""
    if rc < 0:
        raise PangaeaError(rc, load().pg_last_error().decode(errors="replace"))
    return rc
