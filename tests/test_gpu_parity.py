"""Parity of the HIP path (through the C ABI) with the oracle and with the reference's golden outputs.

Bit-exact throughout: these are integer counts.  Run on the GPU box: ``pytest -m gpu``.
"""
import gzip
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from pangaea_amd import _lib, kmer, synth
from pangaea_amd.reads import ReadStream, Rows

from .conftest import GOLDEN

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _CASES = json.load(_f)["cases"]

DEV = "cuda:0"


def _stream(spec):
    r1 = os.path.join(GOLDEN, spec.get("i") or spec["1"])
    r2 = os.path.join(GOLDEN, spec["2"]) if "2" in spec else None
    return ReadStream.from_fastq(r1, r2, device=DEV), oracle.Reads(r1, r2)


def _csv_bytes(tmp_path, names, mat_i32):
    mat = np.ascontiguousarray(mat_i32.cpu().numpy(), dtype=np.int32)
    blob = b"".join(n.encode() + b"\0" for n in names)
    out = str(tmp_path / "o.gz")
    _lib.check(_lib.load().pg_write_csv_gz(out.encode(), blob, mat.ctypes.data, mat.shape[0], mat.shape[1]))
    with gzip.open(out, "rb") as f:
        return f.read()


def test_loaded_library_is_the_in_tree_hip_build():
    assert torch.cuda.is_available()
    assert os.path.dirname(_lib.LIB_PATH).endswith("pangaea_amd")
    assert _lib.load().pg_device_count() >= 1
    # ... and it is the PRODUCT build: no timing experiment, no phase stamps (a checked library only when the run asked for it)
    want = _lib.BUILD_CHECKED if os.environ.get("PANGAEA_LIB", "") == "checked" else 0
    assert _lib.load().pg_build_flags() == want


# ------------------------------------------------------------------ golden vectors of the reference binaries


@pytest.mark.parametrize("case", [c for c in _CASES if c["tool"] == "count_tnf"], ids=lambda c: c["expect"])
def test_tnf_golden(case, tmp_path):
    s, _ = _stream(case["input"])
    rows = s.rows(case["min_len"])
    if case["k"] > _lib.TNF_MAX_K:
        pytest.skip("k_tnf beyond the LDS histogram")
    tnf, _ = kmer.features(s, rows, k_tnf=case["k"])
    with open(os.path.join(GOLDEN, case["expect"]), "rb") as f:
        assert _csv_bytes(tmp_path, rows.names, tnf) == f.read()


@pytest.mark.parametrize("kind", ["default", "hash"])
@pytest.mark.parametrize("case", [c for c in _CASES if c["tool"] == "count_kmer"], ids=lambda c: c["expect"])
def test_abundance_golden(case, kind, tmp_path):
    k = case["k"]
    if k > _lib.HASH_MAX_K:
        if kind == "hash":
            with pytest.raises(ValueError):
                kmer.KmerTable.alloc(k, DEV, "hash")
            return
        assert kmer.KmerTable.default_kind(k) == "wide"
    s, _ = _stream(case["input"])
    rows = s.rows(case["min_len"])
    # the table the reference was given: the (possibly holed) dump, loaded with its own loader semantics
    codes, counts = oracle.Table.from_dump(os.path.join(GOLDEN, case["dump"]), k).items()
    table = kmer.KmerTable.from_items(k, codes, counts, DEV, None if kind == "default" else "hash")
    if table.kind == "hash" and case["window"] * case["vsize"] > _lib.HASH_COUNT_SAT:
        with pytest.raises(_lib.PangaeaError):      # bins beyond the exact range of the slot counter are refused
            kmer.features(s, rows, k_tnf=None, table=table, window=case["window"], vsize=case["vsize"])
        return
    _, abd = kmer.features(s, rows, k_tnf=None, table=table, window=case["window"], vsize=case["vsize"])
    with open(os.path.join(GOLDEN, case["expect"]), "rb") as f:
        assert _csv_bytes(tmp_path, rows.names, abd) == f.read()
    if not case["holes"]:
        # and the table the GPU counts by itself is the dump
        # (cases made under jellyfish's rules: lower-case bases count; the quality threshold of paired files is in the stream)
        mine = kmer.count_kmers(s, k, kind=None if kind == "default" else "hash", lowercase_is_base=case.get("jellyfish_rules", False))
        c2, n2 = mine.items()
        assert np.array_equal(c2, codes) and np.array_equal(n2, np.minimum(counts, _lib.HASH_COUNT_SAT) if mine.kind == "hash" else counts)
        assert mine.kind == ("wide" if k > 21 else "dense" if (kind == "default" and k <= 8) else "hash")


# ------------------------------------------------------------------ seeded synthetic reads vs the oracle


def _oracle_rows(s: ReadStream, rows: Rows, k_tnf, k, window, vsize):
    text = s.decode()
    table = oracle.Table(k, threads=4).count(text) if k else None
    tnf = np.stack([oracle.tnf_row(text[a:b], k_tnf) for a, b in zip(rows.start, rows.end)]) if k_tnf else None
    abd = np.stack([oracle.abd_row(text[a:b], k, table, window, vsize) for a, b in zip(rows.start, rows.end)]) if k else None
    return table, tnf, abd


@pytest.mark.parametrize("k,kind,k_tnf,window,vsize,seg", [
    (15, "dense", 4, 10, 400, 16384), (21, "hash", 4, 10, 400, 16384), (21, "hash", 3, 1, 6, 64),
    (22, "wide", 4, 10, 400, 16384), (27, "wide", 2, 1, 6, 96), (31, "wide", 4, 3, 700, 4096), (13, "wide", 4, 2, 50, 512),
    (11, "dense", 5, 2, 50, 32), (11, "hash", 6, 3, 1500, 4096), (17, "hash", 1, 7, 33, 1024), (4, "dense", 2, 100, 400, 512),
])
def test_synthetic_against_oracle(k, kind, k_tnf, window, vsize, seg):
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=37, n_genomes=3, genome_len=30_000, fragment=8_000,
                            sub_rate=0.01, n_rate=0.2, seed=100 + k)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    assert len(rows) == 37
    table = kmer.count_kmers(s, k, kind=kind)
    tnf, abd = kmer.features(s, rows, k_tnf=k_tnf, table=table, window=window, vsize=vsize, seg_chars=seg)
    otab, otnf, oabd = _oracle_rows(s, rows, k_tnf, k, window, vsize)
    gc, gn = table.items()
    oc, on = otab.items()
    assert np.array_equal(gc, oc) and np.array_equal(gn, on)
    assert np.array_equal(tnf.cpu().numpy(), otnf)
    assert np.array_equal(abd.cpu().numpy(), oabd)


@pytest.mark.parametrize("k,k_tnf,window,vsize,log2_slots,log2_bucket", [
    (21, 4, 10, 400, None, None), (21, 4, 1, 6, 19, 7), (17, 3, 2, 512, 18, 14), (9, 5, 3, 33, 20, 5), (21, None, 10, 400, 18, 10)])
def test_abundance_by_shuffle_equals_lookups_and_oracle(k, k_tnf, window, vsize, log2_slots, log2_bucket):
    """count(stream, rows=plan) keeps row-tagged records; features() then answers from LDS and shuffles (row, bin) words
    back by row group -- same matrices as the lookup kernel and the oracle"""
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=37, n_genomes=3, genome_len=30_000, fragment=8_000,
                            sub_rate=0.01, n_rate=0.2, seed=200 + k)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV, seg_chars=1024)
    if log2_slots is None:
        table = kmer.count_kmers(s, k, kind="hash", rows=plan)
    else:
        table = kmer.KmerTable.with_slots(k, DEV, log2_slots, log2_bucket).count(s, rows=plan)
    assert table.has_records_for(plan, vsize)
    tnf, abd = kmer.features(s, plan, k_tnf=k_tnf, table=table, window=window, vsize=vsize)
    _, otnf, oabd = _oracle_rows(s, rows, k_tnf, k, window, vsize)
    assert np.array_equal(abd.cpu().numpy(), oabd)
    if k_tnf:
        assert np.array_equal(tnf.cpu().numpy(), otnf)
    # the lookup kernel on the same table (a Rows argument has no records attached)
    _, abd2 = kmer.features(s, rows, k_tnf=None, table=table, window=window, vsize=vsize)
    assert torch.equal(abd, abd2)
    # beyond the shuffle's bin range the lookup kernel takes over transparently
    assert not table.has_records_for(plan, 1024)
    _, abd3 = kmer.features(s, plan, k_tnf=None, table=table, window=1, vsize=1024)
    _, _, oabd3 = _oracle_rows(s, rows, None, k, 1, 1024)
    assert np.array_equal(abd3.cpu().numpy(), oabd3)
    # a second count invalidates the records (they describe one pass only)
    table.count(s, rows=plan)
    assert not table.has_records_for(plan, vsize)


def test_shuffle_with_many_small_rows_and_ragged_rows():
    # > 16 384 rows: the (row, bin) words take two scatter passes; rows of 2 pairs, some not word aligned
    cfg = synth.SynthConfig(n_pairs=36_000, n_barcodes=18_000, n_genomes=2, genome_len=40_000, fragment=2_000, unbarcoded=0.0, seed=8)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(300)
    assert len(rows) > 16_384 + 64
    plan = kmer.Plan(rows, DEV)
    table = kmer.count_kmers(s, 15, kind="hash", rows=plan)
    _, abd = kmer.features(s, plan, k_tnf=None, table=table, window=2, vsize=40)
    _, abd2 = kmer.features(s, rows, k_tnf=None, table=table, window=2, vsize=40)
    assert torch.equal(abd, abd2)
    otab = oracle.Table(15, threads=4).count(s.decode())
    for r in (0, 1, 63, 64, 16_383, 16_384, len(rows) - 1):
        txt = s.decode(int(rows.start[r]), int(rows.end[r]))
        assert np.array_equal(abd[r].cpu().numpy(), oracle.abd_row(txt, 15, otab, 2, 40))
    # ragged runs: empty, 1-character and unaligned rows, rows without any k-mer
    runs = [("", b"ACGTACGTN"), ("b", b"N"), ("c", b"ACGN"), ("d", b"ACGTN" * 3), ("e", b"A" * 5000 + b"N"),
            ("f", b"NNNNACGTACGTACGTACGTACGTACGTNNNN"), ("h", (b"ACGT" * 9 + b"N") * 40), ("i", b"ACGTTGCAN" * 7),
            ("j", b"G" * 31 + b"N"), ("k", b"C" * 32 + b"N"), ("l", b"T" * 33 + b"N")]
    s = ReadStream.from_runs(runs, device=DEV)
    rows = s.rows(0)
    plan = kmer.Plan(rows, DEV, seg_chars=32)
    table = kmer.KmerTable.with_slots(9, DEV, 14, 8).count(s, rows=plan)
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=table, window=1, vsize=64)
    _, otnf, oabd = _oracle_rows(s, rows, 4, 9, 1, 64)
    assert np.array_equal(tnf.cpu().numpy(), otnf) and np.array_equal(abd.cpu().numpy(), oabd)


def test_ragged_and_degenerate_runs():
    runs = [("", b"ACGTACGTN"), ("a", b""), ("b", b"N"), ("c", b"ACGN"), ("d", b"ACGTN" * 3), ("e", b"A" * 5000 + b"N"),
            ("f", b"NNNNACGTACGTACGTACGTACGTACGTNNNN"), ("g", b"acgtacgtacgtN"), ("h", (b"ACGT" * 9 + b"N") * 40),
            ("i", b"ACGTTGCAN" * 7), ("j", b"G" * 31 + b"N"), ("k", b"C" * 32 + b"N"), ("l", b"T" * 33 + b"N")]
    s = ReadStream.from_runs(runs, device=DEV)
    for k, kind in ((4, "dense"), (21, "hash"), (9, "hash")):
        rows = s.rows(-1)             # every named run, even the empty one
        assert len(rows) == len(runs) - 1
        table = kmer.count_kmers(s, k, kind=kind)
        tnf, abd = kmer.features(s, rows, k_tnf=4, table=table, window=1, vsize=64, seg_chars=32)
        otab, otnf, oabd = _oracle_rows(s, rows, 4, k, 1, 64)
        assert all(np.array_equal(x, y) for x, y in zip(table.items(), otab.items()))
        assert np.array_equal(tnf.cpu().numpy(), otnf) and np.array_equal(abd.cpu().numpy(), oabd)
    none = Rows(np.zeros(0, np.int64), [], np.zeros(0, np.int64), np.zeros(0, np.int64))
    tnf, _ = kmer.features(s, none, k_tnf=4)
    assert tuple(tnf.shape) == (0, 136)


def test_counting_in_pieces_accumulates():
    cfg = synth.SynthConfig(n_pairs=4096, n_barcodes=16, n_genomes=2, genome_len=50_000, fragment=10_000, seed=5)
    s = synth.generate(cfg, device=DEV)
    for k, kind in ((13, "dense"), (21, "hash")):
        whole = kmer.count_kmers(s, k, kind=kind)
        parts = kmer.KmerTable.alloc(k, DEV, kind, distinct_hint=s.n_chars)
        cut = (s.n_words // 3) + 7
        parts.count(s, 0, cut).count(s, cut, s.n_words)
        assert all(np.array_equal(x, y) for x, y in zip(whole.items(), parts.items()))
        # merging a compacted table doubles every count
        if kind == "hash":
            twice = kmer.KmerTable.alloc(k, DEV, kind, distinct_hint=s.n_chars)
            twice.merge(whole.compact()).merge(whole.compact())
            c1, n1 = whole.items()
            c2, n2 = twice.items()
            assert np.array_equal(c1, c2) and np.array_equal(2 * n1, n2)


@pytest.mark.parametrize("log2_slots,log2_bucket", [(18, 14), (18, 10), (19, 7), (20, 5), (20, 6), (22, 5), (23, 7), (18, 0)])
def test_bucketed_counter_equals_direct_counter(log2_slots, log2_bucket):
    """partition + LDS counting (one or two scatter passes, 2^4..2^15 buckets) builds the same multiset as one global
    atomic per occurrence, fresh and accumulating, and the lookups see the same counts"""
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=20, n_genomes=3, genome_len=20_000, fragment=8_000, n_rate=0.05, seed=31)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    want = oracle.Table(21, threads=4).count(s.decode())
    wc, wn = want.items()
    t = kmer.KmerTable.with_slots(21, DEV, log2_slots, log2_bucket)
    assert t.log2_bucket == log2_bucket
    t.count(s)
    gc, gn = t.items()
    assert np.array_equal(gc, wc) and np.array_equal(gn, wn)
    _, abd = kmer.features(s, rows, k_tnf=None, table=t, window=2, vsize=50)
    _, _, oabd = _oracle_rows(s, rows, None, 21, 2, 50)
    assert np.array_equal(abd.cpu().numpy(), oabd)
    # accumulate: the same stream again in three pieces doubles every count; reset() forgets everything
    a, b = s.n_words // 4 + 3, s.n_words // 2 + 11
    t.count(s, 0, a).count(s, a, b).count(s, b, s.n_words)
    gc, gn = t.items()
    assert np.array_equal(gc, wc) and np.array_equal(gn, 2 * wn)
    t.reset().count(s)
    gc, gn = t.items()
    assert np.array_equal(gc, wc) and np.array_equal(gn, wn)


@pytest.mark.parametrize("log2_slots,log2_bucket", [(18, 10), (20, 6), (18, 14), (18, 0), (22, 10), (25, 14)])
def test_merging_tables_bucket_by_bucket(log2_slots, log2_bucket):
    """tables of two halves of a stream (same geometry) merge into the table of the whole, through the LDS bucket merge
    (or global atomics for the unbucketed form) -- what the multi-GPU exchange does after its all-gather"""
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=20, n_genomes=3, genome_len=20_000, fragment=8_000, seed=41)
    s = synth.generate(cfg, device=DEV)
    cut = s.n_words // 2 + 5
    a = kmer.KmerTable.with_slots(21, DEV, log2_slots, log2_bucket).count(s, 0, cut)
    b = kmer.KmerTable.with_slots(21, DEV, log2_slots, log2_bucket).count(s, cut, s.n_words)
    c = kmer.KmerTable.with_slots(21, DEV, log2_slots, log2_bucket).count(s, cut, s.n_words)
    assert int(b.bucket_counts().sum()) == b.compact().numel() and b.bucket_counts().numel() == b.n_buckets
    a.merge_parts([(b.compact(), b.bucket_counts()), (c.compact(), c.bucket_counts())])
    want = oracle.Table(21, threads=4).count(s.decode())
    gc, gn = a.items()
    # a + 2 x b: every k-mer of the second half counted twice more
    wb = kmer.KmerTable.with_slots(21, DEV, log2_slots, 0).count(s).count(s, cut, s.n_words)
    assert all(np.array_equal(x, y) for x, y in zip((gc, gn), wb.items()))
    assert len(gc) == len(want.items()[0])


@pytest.mark.parametrize("log2_slots,log2_bucket", [(18, 10), (20, 6), (18, 14), (17, 5), (22, 10), (25, 14)])
def test_exchange_kernels_fill_compact_rebuild(log2_slots, log2_bucket):
    """the three launches of the multi-GPU exchange: per-bucket fills, bucket-ordered compaction into a padded gather
    buffer, and the rebuild of the table from all parts (own one included, old slots ignored)"""
    cfg = synth.SynthConfig(n_pairs=2500 if log2_slots > 17 else 60, n_barcodes=20, n_genomes=3, genome_len=20_000, fragment=8_000, seed=43)
    s = synth.generate(cfg, device=DEV)
    cut = s.n_words // 3 + 2
    parts = [kmer.KmerTable.with_slots(21, DEV, log2_slots, log2_bucket).count(s, a, b)
             for a, b in ((0, cut), (cut, s.n_words), (cut, s.n_words))]
    fills = torch.stack([t.bucket_fill() for t in parts])
    assert all(torch.equal(f, t.bucket_counts()) for f, t in zip(fills, parts))
    nb, world = parts[0].n_buckets, len(parts)
    cap = int(fills.sum(1).max()) + 7                           # padding of the gather buffer is never read
    seg = torch.zeros((world, nb + 1), dtype=torch.int64, device=DEV)
    seg[:, 1:] = torch.cumsum(fills, 1)
    buf = torch.full((world * cap,), -1, dtype=torch.int64, device=DEV)
    for r, t in enumerate(parts):
        t.compact_into(buf[r * cap:(r + 1) * cap], seg[r].contiguous())
        got = buf[r * cap:r * cap + int(fills[r].sum())]
        assert torch.equal(torch.sort(got).values, torch.sort(t.compact()).values)
        # bucket by bucket: the segment of bucket b holds exactly that bucket's occupied slots
        b = nb // 2
        seg_b = got[int(seg[r, b]):int(seg[r, b + 1])]
        assert torch.equal(torch.sort(seg_b).values, torch.sort(t.data.view(nb, -1)[b][t.data.view(nb, -1)[b] != 0]).values)
    seg += torch.arange(world, device=DEV)[:, None] * cap
    out = kmer.KmerTable.with_slots(21, DEV, log2_slots, log2_bucket)
    out.data.fill_(0x7FFF_FFFF_FFFF)                            # garbage: a rebuild must not read the old slots
    out.rebuild_from(buf, seg)
    want = kmer.KmerTable.with_slots(21, DEV, log2_slots, 0).count(s).count(s, cut, s.n_words)
    assert all(np.array_equal(x, y) for x, y in zip(out.items(), want.items()))
    # same result as the accumulate-into-own-table form
    parts[0].merge_parts([(t.compact(), t.bucket_counts()) for t in parts[1:]])
    assert all(np.array_equal(x, y) for x, y in zip(out.items(), parts[0].items()))


@pytest.mark.parametrize("log2_slots,log2_bucket,g", [(20, 10, 0), (20, 10, 1), (21, 9, 3), (22, 12, 2), (19, 10, 1), (22, 10, 1), (23, 10, 2), (25, 14, 0), (25, 12, 3)])
def test_deferred_count_gather_rebuild_equals_direct_count(log2_slots, log2_bucket, g):
    """multi-GPU counting form: the table (union geometry) is not written; groups of 2^g buckets are counted in LDS, the
    entries are gathered bucket by bucket and the table is rebuilt from them -- same table as counting into it directly,
    and the row-tagged records left behind still answer the abundance rows by shuffle"""
    cfg = synth.SynthConfig(n_pairs=2500, n_barcodes=20, n_genomes=3, genome_len=20_000, fragment=8_000, n_rate=0.05, seed=47)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    want = kmer.KmerTable.with_slots(21, DEV, log2_slots, log2_bucket).count(s)
    t = kmer.KmerTable.with_slots(21, DEV, log2_slots, log2_bucket)
    t.data.fill_(0x7FFF_FFFF_FFFF)                               # the slots are neither read nor written until the rebuild
    t._empty = True
    assert t.can_defer(s.n_words)
    t.count(s, rows=plan, deferred_group=g)
    assert t.pending and int((t.data != 0x7FFF_FFFF_FFFF).sum()) == 0
    with pytest.raises(RuntimeError, match="deferred"):
        t.items()
    with pytest.raises(RuntimeError, match="deferred"):
        kmer.features(s, plan, k_tnf=None, table=t)
    fill = t.deferred_fill()
    assert torch.equal(fill, want.bucket_counts())
    seg = torch.zeros((2, t.n_buckets + 1), dtype=torch.int64, device=DEV)
    seg[:, 1:] = torch.cumsum(fill, 0)
    cap = int(fill.sum()) + 3
    buf = torch.full((2 * cap,), -1, dtype=torch.int64, device=DEV)
    t.deferred_compact_into(buf[:cap], seg[0].contiguous())
    assert torch.equal(torch.sort(buf[:cap - 3]).values, torch.sort(want.compact()).values)
    # "two ranks" with the same shard: every count doubles
    buf[cap:2 * cap - 3] = buf[:cap - 3]
    seg[1] += cap
    t.rebuild_from(buf, seg)
    assert not t.pending
    wc, wn = want.items()
    gc, gn = t.items()
    assert np.array_equal(gc, wc) and np.array_equal(gn, 2 * wn)
    # K3 by shuffle from the records of the deferred count, against the rebuilt table
    assert t.has_records_for(plan, 64)
    _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=2, vsize=64)
    _, abd2 = kmer.features(s, rows, k_tnf=None, table=t, window=2, vsize=64)
    assert torch.equal(abd, abd2)
    doubled = kmer.KmerTable.with_slots(21, DEV, log2_slots, 0).count(s).count(s)
    _, abd3 = kmer.features(s, rows, k_tnf=None, table=doubled, window=2, vsize=64)
    assert torch.equal(abd, abd3)


def test_deferred_count_reports_a_full_group_table():
    cfg = synth.SynthConfig(n_pairs=4000, n_barcodes=16, n_genomes=2, genome_len=300_000, fragment=60_000, seed=77)
    s = synth.generate(cfg, device=DEV)
    t = kmer.KmerTable.with_slots(21, DEV, 21, 9)              # fine as a table, but 8 buckets' keys do not fit one LDS table
    t.count(s)
    t.reset()
    with pytest.raises(_lib.PangaeaError) as e:
        t.count(s, deferred_group=3)
    assert e.value.code == _lib.PG_ETABLEFULL


def test_bucket_overflow_is_reported():
    cfg = synth.SynthConfig(n_pairs=4000, n_barcodes=16, n_genomes=2, genome_len=300_000, fragment=60_000, seed=77)
    s = synth.generate(cfg, device=DEV)
    t = kmer.KmerTable.with_slots(21, DEV, 14, 6)          # 16 k slots for ~1 M distinct 21-mers
    with pytest.raises(_lib.PangaeaError) as e:
        t.count(s)
    assert e.value.code == _lib.PG_ETABLEFULL
    grown = kmer.count_kmers(s, 21, kind="hash", distinct_hint=4096, log2_bucket=6)
    want = oracle.Table(21, threads=4).count(s.decode())
    assert all(np.array_equal(x, y) for x, y in zip(grown.items(), want.items()))


def test_hash_table_full_is_reported_and_regrown():
    cfg = synth.SynthConfig(n_pairs=2048, n_barcodes=8, n_genomes=2, genome_len=200_000, fragment=50_000, seed=9)
    s = synth.generate(cfg, device=DEV)
    tiny = kmer.KmerTable.alloc(21, DEV, "hash", distinct_hint=512)
    with pytest.raises(_lib.PangaeaError) as e:
        tiny.count(s)
    assert e.value.code == _lib.PG_ETABLEFULL
    grown = kmer.count_kmers(s, 21, kind="hash", distinct_hint=512)
    want = oracle.Table(21, threads=4).count(s.decode())
    assert all(np.array_equal(x, y) for x, y in zip(grown.items(), want.items()))


def test_saturating_counts_do_not_change_bins():
    # 3.1 M identical 21-mers: the slot count stops at 2^21, every bin below vsize*window stays exact
    s = ReadStream.from_runs([("a", b"A" * 1_600_000 + b"N" + b"T" * 1_500_040 + b"N"), ("b", b"ACGT" * 600 + b"N")], device=DEV)
    table = kmer.count_kmers(s, 21, kind="hash")
    codes, counts = table.items()
    assert counts.max() >= _lib.HASH_COUNT_SAT and counts.max() < (1 << _lib.HASH_COUNT_BITS)
    rows = s.rows(0)
    _, abd = kmer.features(s, rows, k_tnf=None, table=table, window=10, vsize=400)
    _, _, oabd = _oracle_rows(s, rows, None, 21, 10, 400)
    assert np.array_equal(abd.cpu().numpy(), oabd)
    with pytest.raises(_lib.PangaeaError):
        kmer.features(s, rows, k_tnf=None, table=table, window=1 << 12, vsize=1 << 10)


# ------------------------------------------------------------------ BASELINE-size properties (config 2)


def _valid_kmer_ends(valid_words: torch.Tensor, k: int) -> torch.Tensor:
    """per word: number of positions ending a run of >= k valid characters (torch restatement)"""
    v = valid_words.to(torch.int64) & 0xFFFFFFFF
    prev = torch.cat([v.new_zeros(1), v[:-1]])
    m = (v << 32) | prev                     # bit 63 may be set: arithmetic below only uses & and <<
    r = m
    ln = 1
    while 2 * ln <= k:
        r = r & (r << ln)
        ln *= 2
    if ln < k:
        r = r & (r << (k - ln))
    hi = (r >> 32) & 0xFFFFFFFF
    cnt = torch.zeros_like(hi)
    for b in range(32):
        cnt += (hi >> b) & 1
    return cnt


def test_full_size_invariants():
    """10 M pairs / 50 k barcodes / k=21 (BASELINE config 2): size-independent properties"""
    cfg = synth.SynthConfig(n_pairs=10_000_000, n_barcodes=50_000)
    s = synth.generate(cfg, device=DEV, chunk_pairs=1 << 17, with_names=False)
    rows = s.rows(2000)
    assert len(rows) == 50_000
    plan = kmer.Plan(rows, DEV)
    table = kmer.count_kmers(s, 21, kind="hash", distinct_hint=260_000_000, rows=plan)
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=table, window=10, vsize=400)          # shuffle path
    tnf_l, abd_l = kmer.features(s, rows, k_tnf=4, table=table, window=10, vsize=400)      # lookup path
    assert torch.equal(tnf, tnf_l) and torch.equal(abd, abd_l)
    torch.cuda.synchronize()
    n21 = _valid_kmer_ends(s.valid, 21)
    n4 = _valid_kmer_ends(s.valid, 4)
    # (1) the table accounts for every valid 21-mer occurrence of every read
    slots = table.compact()
    assert int((slots & ((1 << 22) - 1)).sum().item()) == int(n21.sum().item())
    # (2) TNF rows + the two complement ranges (first pair, unbarcoded tail) account for every valid 4-mer end;
    #     rows tile [302, end of the last barcoded run) without gaps
    cpp = cfg.chars_per_pair
    assert rows.start[0] == cpp and (np.diff(rows.start) == cfg.pairs_per_barcode * cpp).all() and (rows.end[:-1] == rows.start[1:]).all()
    comp = Rows(np.arange(2), ["head", "tail"], np.array([0, int(rows.end[-1])]), np.array([cpp, s.n_chars]))
    tnf_c, abd_c = kmer.features(s, comp, k_tnf=4, table=table, window=1 << 11, vsize=1 << 10)
    assert int(tnf.to(torch.int64).sum().item()) + int(tnf_c.sum().item()) == int(n4.sum().item())
    # (3) abundance rows: every valid 21-mer of a row is binned unless its multiplicity is >= 4000
    #     (window * vsize = 2^21 = the saturation point, far above any multiplicity here: nothing is dropped)
    _, abd_all = kmer.features(s, rows, k_tnf=None, table=table, window=1 << 11, vsize=1 << 10)
    assert int(abd_all.to(torch.int64).sum().item()) + int(abd_c.sum().item()) == int(n21.sum().item())
    assert (abd.to(torch.int64).sum(1) <= abd_all.to(torch.int64).sum(1)).all()
    # (4) idempotence: a second launch into fresh outputs gives the same matrices
    tnf2, abd2 = kmer.features(s, plan, k_tnf=4, table=table, window=10, vsize=400)
    assert torch.equal(tnf, tnf2) and torch.equal(abd, abd2)
    # (5) spot rows against the oracle (first, middle, last)
    text_rows = [0, 24_999, 49_999]
    for r in text_rows:
        txt = s.decode(int(rows.start[r]), int(rows.end[r]))
        assert np.array_equal(tnf[r].cpu().numpy(), oracle.tnf_row(txt, 4))


def test_distinct_estimate_sizes_tables():
    """the HyperLogLog pass lands within a few percent of the true number of distinct canonical k-mers, at both ends of
    its range, and count_kmers sizes its table from it (no overflow-and-redo on low-coverage input)"""
    for n_pairs, genome_len in ((300, 1000), (20_000, 400_000)):
        cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=max(2, n_pairs // 150), n_genomes=3, genome_len=genome_len,
                                fragment=min(10_000, genome_len // 2), seed=13)
        s = synth.generate(cfg, device=DEV)
        for k in (11, 21, 31):
            true = len(oracle.Table(k, threads=4).count(s.decode()))
            est = kmer.estimate_distinct(s, k)
            assert abs(est - true) <= 0.06 * true + 20, (n_pairs, k, est, true)
            t = kmer.count_kmers(s, k)
            assert len(t.items()[0]) == true
            assert (1 << t.log2_slots) >= 2 * true and (1 << t.log2_slots) <= max(1 << 16, 12 * true)


@pytest.mark.parametrize("seed", range(12))
def test_randomized_configurations(seed):
    """random k / bucket geometry / window / vector size / row filter / error rates: partition + shuffle path == lookup path
    == oracle, bit for bit"""
    rng = np.random.RandomState(1000 + seed)
    k = int(rng.choice([5, 9, 12, 15, 16, 19, 21]))
    k_tnf = int(rng.choice([1, 2, 3, 4, 5, 6]))
    window = int(rng.choice([1, 2, 7, 10, 25]))
    vsize = int(rng.choice([3, 17, 64, 400, 512]))
    n_pairs = int(rng.choice([64, 700, 2500]))
    n_bc = int(rng.choice([1, 7, 40]))
    cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=min(n_bc, n_pairs // 2), n_genomes=int(rng.randint(1, 4)),
                            genome_len=int(rng.choice([3_000, 50_000])), fragment=2_000, sub_rate=float(rng.choice([0.0, 0.02])),
                            n_rate=float(rng.choice([0.0, 0.3, 1.0])), unbarcoded=float(rng.choice([0.0, 0.1])), seed=seed)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(int(rng.choice([0, 302, 2000])))
    plan = kmer.Plan(rows, DEV, seg_chars=int(rng.choice([32, 256, 16384])))
    log2_slots = int(rng.choice([17, 18, 20]))
    log2_bucket = int(rng.choice([5, 8, 11, 14]))
    table = kmer.KmerTable.with_slots(k, DEV, log2_slots, log2_bucket).count(s, rows=plan)
    tnf, abd = kmer.features(s, plan, k_tnf=k_tnf, table=table, window=window, vsize=vsize)
    tnf_l, abd_l = kmer.features(s, rows, k_tnf=k_tnf, table=table, window=window, vsize=vsize, seg_chars=64)
    assert torch.equal(tnf, tnf_l) and torch.equal(abd, abd_l)
    # the fused count + lookup form (applies from 2^11 buckets on, silently ignored otherwise): same table, same rows
    fused = kmer.KmerTable.with_slots(k, DEV, log2_slots, log2_bucket).count(s, rows=plan, emit=(window, vsize))
    assert (fused._emitted is not None) == (len(rows) > 0 and log2_slots - log2_bucket >= 11)
    _, abd_f = kmer.features(s, plan, k_tnf=None, table=fused, window=window, vsize=vsize)
    assert torch.equal(abd_f, abd) and all(np.array_equal(x, y) for x, y in zip(fused.items(), table.items()))   # (slot order is free)
    if len(rows):
        otab, otnf, oabd = _oracle_rows(s, rows, k_tnf, k, window, vsize)
        assert all(np.array_equal(x, y) for x, y in zip(table.items(), otab.items()))
        assert np.array_equal(tnf.cpu().numpy(), otnf) and np.array_equal(abd.cpu().numpy(), oabd)
    else:
        assert tuple(abd.shape) == (0, vsize)


@pytest.mark.parametrize("n_barcodes,force_two", [(70_000, False), (40_000, True), (40_000, False)])
def test_row_shuffle_one_pass_and_two_pass_forms(n_barcodes, force_two, monkeypatch):
    """up to 65536 rows the (row, bin) words go to their row groups in one 1024-way pass, beyond that in two; both forms
    (the second also forced on a smaller row set) give the lookup kernel's matrix"""
    if force_two:
        monkeypatch.setenv("PG_S2_TWO_PASS", "1")
    cfg = synth.SynthConfig(n_pairs=4 * n_barcodes, n_barcodes=n_barcodes, n_genomes=4, genome_len=200_000, fragment=20_000, unbarcoded=0.0, seed=61)
    s = synth.generate(cfg, device=DEV, with_names=True)
    rows = s.rows(600)
    assert len(rows) > (65536 if n_barcodes > 65536 else 32768)
    plan = kmer.Plan(rows, DEV)
    table = kmer.count_kmers(s, 21, rows=plan)
    assert table.has_records_for(plan, 400)
    _, abd = kmer.features(s, plan, k_tnf=None, table=table, window=3, vsize=400)
    _, abd2 = kmer.features(s, rows, k_tnf=None, table=table, window=3, vsize=400)
    assert torch.equal(abd, abd2) and int(abd.sum()) > 0


def test_mid_scale_table_and_rows_against_oracle():
    """200 k pairs (52 M k-mer occurrences, 2^15-bucket geometry with the split 32-bit LDS tables, one-pass row shuffle): the
    whole multiplicity table equals the oracle's, and so do the TNF / abundance rows of a spread of barcodes"""
    cfg = synth.SynthConfig(n_pairs=200_000, n_barcodes=1000, n_genomes=8, genome_len=400_000, fragment=40_000, sub_rate=0.002, n_rate=0.01, seed=71)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    table = kmer.KmerTable.with_slots(21, DEV, 29, 14).count(s, rows=plan)         # the bench's geometry
    assert table.tag_bits <= 31 and table.has_records_for(plan, 400)
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=table, window=10, vsize=400)
    text = s.decode()
    otab = oracle.Table(21, threads=8).count(text)
    assert all(np.array_equal(x, y) for x, y in zip(table.items(), otab.items()))
    pick = list(range(0, len(rows), max(1, len(rows) // 25)))
    for i in pick:
        seq = text[int(rows.start[i]):int(rows.end[i])]
        assert np.array_equal(tnf[i].cpu().numpy(), oracle.tnf_row(seq, 4))
        assert np.array_equal(abd[i].cpu().numpy(), oracle.abd_row(seq, 21, otab, 10, 400))


@pytest.mark.parametrize("k,window,vsize,log2_slots,log2_bucket", [(21, 10, 400, 22, 10), (15, 3, 64, 24, 13), (21, 1, 512, 29, 14), (9, 2, 33, 21, 10)])
def test_fused_count_and_lookup_equals_separate_kernels(k, window, vsize, log2_slots, log2_bucket):
    """count(emit=(window, vsize)): the abundance lookups ride inside the counting kernel; table and matrix are those of the
    separate kernels and of the oracle, a second features() call falls back to the table, other parameters too"""
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=37, n_genomes=3, genome_len=30_000, fragment=8_000, sub_rate=0.01, n_rate=0.2, seed=300 + k)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    plain = kmer.KmerTable.with_slots(k, DEV, log2_slots, log2_bucket).count(s, rows=plan)
    _, want = kmer.features(s, plan, k_tnf=None, table=plain, window=window, vsize=vsize)
    fused = kmer.KmerTable.with_slots(k, DEV, log2_slots, log2_bucket)
    fused.data.fill_(0x7FFF_FFFF_FFFF); fused._empty = True
    fused.count(s, rows=plan, emit=(window, vsize))
    assert fused._emitted == (window, vsize)
    assert all(np.array_equal(x, y) for x, y in zip(fused.items(), plain.items()))
    _, got = kmer.features(s, plan, k_tnf=None, table=fused, window=window, vsize=vsize)
    assert fused._emitted is None and torch.equal(got, want)
    _, oabd = _oracle_rows(s, rows, None, k, window, vsize)[1:] if False else (None, _oracle_rows(s, rows, None, k, window, vsize)[2])
    assert np.array_equal(got.cpu().numpy(), oabd)
    _, again = kmer.features(s, plan, k_tnf=None, table=fused, window=window, vsize=vsize)          # from the records + table now
    assert torch.equal(again, want)
    fused.reset().count(s, rows=plan, emit=(window, vsize))
    _, other = kmer.features(s, plan, k_tnf=None, table=fused, window=window + 1, vsize=vsize)       # other parameters: unfused path
    _, other_want = kmer.features(s, rows, k_tnf=None, table=plain, window=window + 1, vsize=vsize)
    assert torch.equal(other, other_want)
    # where fusion does not apply (few buckets) the argument is ignored
    small = kmer.KmerTable.with_slots(k, DEV, 18, 14).count(s, rows=plan, emit=(window, vsize))
    assert small._emitted is None


@pytest.mark.parametrize("log2_slots,log2_bucket,mode", [(22, 10, "plain"), (22, 10, "emit"), (22, 10, "deferred"), (18, 14, "plain"), (18, 0, "plain")])
def test_lowercase_is_base_counts_like_jellyfish_and_rows_stay_strict(log2_slots, log2_bucket, mode):
    """count(lowercase_is_base=True): the table is jellyfish's (lower-case a c g t are bases: the oracle's counter on the
    upper-cased text), while the TNF / abundance rows keep the reference's own rule (its counters reset on lower case), in
    the direct, bucketed, fused and deferred forms, by shuffle and by lookups"""
    rng = np.random.RandomState(11)
    runs = []
    for b in range(12):
        seq = bytearray(rng.choice(list(b"ACGT"), size=rng.randint(2500, 6000)).astype(np.uint8).tobytes())
        for _ in range(rng.randint(0, 6)):                         # soft-masked stretches, a few N
            a = rng.randint(0, len(seq)); e = min(len(seq), a + rng.randint(1, 80))
            seq[a:e] = bytes(seq[a:e]).lower()
        for _ in range(3):
            seq[rng.randint(0, len(seq))] = ord("N")
        runs.append((f"bc{b}", bytes(seq) + b"N"))
    s = ReadStream.from_runs(runs, device=DEV)
    assert s.valid_lower is not None
    text = b"".join(t for _, t in runs)
    k, window, vsize = 21, 1, 64
    lenient = oracle.Table(k, threads=2).count(text.upper())
    strict = oracle.Table(k, threads=2).count(text)
    assert len(lenient.items()[0]) > len(strict.items()[0])
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    t = kmer.KmerTable.with_slots(k, DEV, log2_slots, log2_bucket)
    if mode == "deferred":
        t.count(s, rows=plan, deferred_group=1, lowercase_is_base=True)
        fill = t.deferred_fill()
        seg = torch.zeros((1, t.n_buckets + 1), dtype=torch.int64, device=DEV)
        seg[0, 1:] = torch.cumsum(fill, 0)
        buf = torch.empty(int(fill.sum()) + 1, dtype=torch.int64, device=DEV)
        t.deferred_compact_into(buf, seg[0].contiguous())
        t.rebuild_from(buf, seg)
    else:
        t.count(s, rows=plan, lowercase_is_base=True, emit=(window, vsize) if mode == "emit" else None)
    assert all(np.array_equal(x, y) for x, y in zip(t.items(), lenient.items()))
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=window, vsize=vsize)
    tnf2, abd2 = kmer.features(s, rows, k_tnf=4, table=t, window=window, vsize=vsize)           # lookup kernel: strict by construction
    assert torch.equal(tnf, tnf2) and torch.equal(abd, abd2)
    for i, (a, b) in enumerate(zip(rows.start, rows.end)):
        seq = text[int(a):int(b)]
        assert np.array_equal(tnf[i].cpu().numpy(), oracle.tnf_row(seq, 4))
        assert np.array_equal(abd[i].cpu().numpy(), oracle.abd_row(seq, k, lenient, window, vsize))
    # default: the strict table
    t2 = kmer.KmerTable.with_slots(k, DEV, log2_slots, log2_bucket).count(s, rows=plan)
    assert all(np.array_equal(x, y) for x, y in zip(t2.items(), strict.items()))
