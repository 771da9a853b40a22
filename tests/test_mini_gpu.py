"""The super-k-mer pipeline (PG_TABLE_MINI: pg_mini_plan + pg_mini_count) against the oracle and against the other HIP paths.

Bit-exact throughout (integer counts).  The table must be the oracle's exact canonical k-mer multiplicities
(jellyfish count -C, src/feature.py:94) and the abundance rows those of count_kmer.cpp:55-108, whatever the minimizer
buckets, record boundaries and round / window structure of the kernels are.
"""
import numpy as np
import pytest
import torch

from oracle import oracle
from pangaea_amd import _lib, kmer, synth
from pangaea_amd.reads import ReadStream, Rows

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle(s, rows, k, window, vsize, k_tnf=4):
    text = s.decode()
    table = oracle.Table(k, threads=4).count(text)
    tnf = np.stack([oracle.tnf_row(text[a:b], k_tnf) for a, b in zip(rows.start, rows.end)]) if len(rows) else None
    abd = np.stack([oracle.abd_row(text[a:b], k, table, window, vsize) for a, b in zip(rows.start, rows.end)]) if len(rows) else None
    return table, tnf, abd


def _same_items(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("k,log2_slots,log2_bucket,window,vsize,min_len,n_pairs", [
    (21, 19, 14, 10, 400, 2000, 3000),      # 32 buckets: first pass only
    (21, 22, 10, 10, 400, 2000, 3000),      # 4096 buckets: both passes
    (21, 18, 14, 1, 6, 302, 800),           # 16 buckets
    (21, 14, 14, 3, 64, 0, 40),             # one bucket, every run a row
    (16, 20, 9, 2, 50, 2000, 3000),         # the smallest k with 13-mer minimizers: windows of 4
    (15, 20, 10, 10, 400, 2000, 3000),      # Pangaea's default k: 11-mer minimizers, windows of 5
    (14, 19, 9, 3, 64, 600, 3000),
    (13, 18, 12, 1, 512, 2000, 3000),       # the smallest k: windows of 3
    (17, 20, 8, 7, 33, 600, 3000),
    (19, 21, 12, 25, 512, 2000, 3000),
    (20, 20, 11, 1, 512, 2000, 3000),
    (21, 24, 8, 10, 400, 2000, 3000),       # 2^16 buckets: 512 regions in the first pass, the plan in two bucket ranges
    (29, 24, 8, 10, 400, 2000, 1500),
    # k > 21 (PG_TABLE_MINI_WIDE: keys + counts planes, the minimizer over the central 8 / 9 M-mers, delayed by 1..5 characters)
    (22, 20, 13, 10, 400, 2000, 3000),
    (23, 19, 10, 3, 64, 600, 1500),
    (24, 20, 12, 2, 50, 2000, 1500),
    (25, 18, 13, 1, 512, 2000, 1500),
    (26, 19, 9, 10, 400, 2000, 1500),
    (27, 20, 11, 10, 400, 2000, 3000),
    (28, 18, 12, 5, 100, 302, 1500),
    (29, 19, 13, 10, 400, 2000, 1500),
    (30, 20, 8, 1, 6, 2000, 1500),
    (31, 20, 13, 10, 400, 2000, 3000),
])
def test_mini_table_and_rows_against_oracle(k, log2_slots, log2_bucket, window, vsize, min_len, n_pairs):
    cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=min(37, max(1, n_pairs // 20)), n_genomes=3, genome_len=30_000, fragment=8_000,
                            sub_rate=0.01, n_rate=0.2, seed=500 + k)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(min_len)
    plan = kmer.Plan(rows, DEV)
    t = kmer.KmerTable.mini_with_slots(k, DEV, log2_slots, log2_bucket)
    t.data.fill_(0x7FFF_FFFF_FFFF)                      # a fresh table is never cleared: every slot must be overwritten
    t.count(s, rows=plan, emit=(window, vsize))
    assert t.kind == ("mini" if k <= 21 else "miniw") and t._emitted == (window, vsize)
    otab, otnf, oabd = _oracle(s, rows, k, window, vsize)
    assert _same_items(t.items(), otab.items())
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=window, vsize=vsize)                 # from the emitted words
    assert t._emitted is None
    assert np.array_equal(tnf.cpu().numpy(), otnf) and np.array_equal(abd.cpu().numpy(), oabd)
    _, abd_l = kmer.features(s, rows, k_tnf=None, table=t, window=window, vsize=vsize, seg_chars=64)  # by lookups in the mini table
    assert torch.equal(abd_l, abd)
    _, abd_o = kmer.features(s, plan, k_tnf=None, table=t, window=window + 1, vsize=vsize)          # other parameters: lookups too
    h = kmer.count_kmers(s, k, kind="hash" if k <= 21 else "wide")
    _, want = kmer.features(s, rows, k_tnf=None, table=h, window=window + 1, vsize=vsize)
    assert torch.equal(abd_o, want)
    # counting again (reset) reuses the partition plan and gives the same table and rows
    plan_ws = t._mini_plan[1]
    t.reset().count(s, rows=plan, emit=(window, vsize))
    assert t._mini_plan[1] is plan_ws
    _, abd2 = kmer.features(s, plan, k_tnf=None, table=t, window=window, vsize=vsize)
    assert torch.equal(abd2, abd) and _same_items(t.items(), otab.items())


@pytest.mark.parametrize("log2_slots,log2_bucket", [(24, 13), (24, 9)])
def test_mini_two_workgroups_per_cu_form_equals_the_one_workgroup_form(log2_slots, log2_bucket, monkeypatch):
    """buckets of at most 2^13 slots are counted by 512-thread workgroups (two per CU, 1024 row-group digits on 512 threads);
    PG_COUNT_BLOCK=1024 runs the same table through the 1024-thread form: same table, same rows, and the oracle's"""
    cfg = synth.SynthConfig(n_pairs=60_000, n_barcodes=700, n_genomes=5, genome_len=60_000, fragment=20_000, sub_rate=0.01, n_rate=0.1, seed=77)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    got = []
    for form in (None, "1024"):
        if form:
            monkeypatch.setenv("PG_COUNT_BLOCK", form)
        t = kmer.KmerTable.mini_with_slots(21, DEV, log2_slots, log2_bucket)
        t.count(s, rows=plan, emit=(10, 400))
        _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)
        got.append((t.items(), abd))
    assert _same_items(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    otab = oracle.Table(21, threads=4).count(s.decode())
    assert _same_items(got[0][0], otab.items())
    text = s.decode()
    for r in range(0, len(rows), max(1, len(rows) // 12)):
        assert np.array_equal(got[0][1][r].cpu().numpy(), oracle.abd_row(text[rows.start[r]:rows.end[r]], 21, otab, 10, 400))


@pytest.mark.parametrize("k,log2_slots,log2_bucket", [(21, 22, 10), (18, 19, 14), (27, 20, 11), (31, 19, 13)])
def test_mini_general_lookup_form(k, log2_slots, log2_bucket, monkeypatch):
    """the lookups' general form (records read and probed a second time: what row sets too large for the (row, slot) words
    take) gives the same rows as the slot form"""
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=37, n_genomes=3, genome_len=30_000, fragment=8_000, sub_rate=0.01, n_rate=0.2, seed=600 + k)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    t = kmer.KmerTable.mini_with_slots(k, DEV, log2_slots, log2_bucket).count(s, rows=plan, emit=(3, 200))
    _, want = kmer.features(s, plan, k_tnf=None, table=t, window=3, vsize=200)
    monkeypatch.setenv("PG_MINI_PROBE_TWICE", "1")
    u = kmer.KmerTable.mini_with_slots(k, DEV, log2_slots, log2_bucket).count(s, rows=plan, emit=(3, 200))
    _, got = kmer.features(s, plan, k_tnf=None, table=u, window=3, vsize=200)
    assert torch.equal(got, want) and _same_items(t.items(), u.items())
    _, _, oabd = _oracle(s, rows, k, 3, 200)
    assert np.array_equal(got.cpu().numpy(), oabd)


def test_mini_table_only_and_from_items():
    cfg = synth.SynthConfig(n_pairs=1500, n_barcodes=11, n_genomes=2, genome_len=20_000, fragment=5_000, sub_rate=0.02, n_rate=0.3, seed=77)
    s = synth.generate(cfg, device=DEV)
    t = kmer.count_kmers(s, 21, kind="mini")                       # no rows, no lookups
    otab = oracle.Table(21, threads=4).count(s.decode())
    assert t.kind == "mini" and _same_items(t.items(), otab.items())
    codes, counts = otab.items()
    u = kmer.KmerTable.from_items(21, codes, counts, DEV, "mini")   # entries of a dump, placed by pg_kmer_merge
    assert _same_items(u.items(), otab.items())
    rows = s.rows(2000)
    _, a = kmer.features(s, rows, k_tnf=None, table=u, window=2, vsize=100)
    _, b = kmer.features(s, rows, k_tnf=None, table=t, window=2, vsize=100)
    assert torch.equal(a, b) and int(a.sum()) > 0
    with pytest.raises(ValueError):
        t.count(s)                                                  # one count per fresh table
    with pytest.raises(ValueError):
        kmer.KmerTable.mini_with_slots(12, DEV, 20)
    # k = 27: entries of a dump into a wide mini table (pg_kmer_merge_wide places them by minimizer bucket), auto-selection
    w = kmer.count_kmers(s, 27, rows=kmer.Plan(rows, DEV), emit=(2, 100))
    otab = oracle.Table(27, threads=4).count(s.decode())
    assert w.kind == "miniw" and _same_items(w.items(), otab.items())
    codes, counts = otab.items()
    v = kmer.KmerTable.from_items(27, codes, counts, DEV, "miniw")
    assert _same_items(v.items(), otab.items())
    _, a = kmer.features(s, rows, k_tnf=None, table=v, window=2, vsize=100)
    _, b = kmer.features(s, rows, k_tnf=None, table=kmer.count_kmers(s, 27, kind="wide"), window=2, vsize=100)
    assert torch.equal(a, b) and int(a.sum()) > 0
    with pytest.raises(ValueError):
        kmer.KmerTable.mini_with_slots(27, DEV, 20, 14)             # 12-byte slots: buckets of at most 2^13


def test_mini_every_kmer_its_own_record():
    """rows of one character cut every record to a single k-mer: 32 records per word, rounds that do not fit the LDS stage and
    are laid out in four windows; the rows are those of the lookup kernel on a key-partitioned table"""
    rng = np.random.RandomState(5)
    n = 40_000
    text = bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
    text = text[:7000] + b"N" + text[7001:19000] + b"NN" + text[19002:]
    s = ReadStream.from_runs([("x", text)], device=DEV)
    start = np.arange(100, n - 100, dtype=np.int64)
    rows = Rows(np.zeros(len(start), dtype=np.int64), [f"r{i}" for i in range(len(start))], start, start + 1)
    plan = kmer.Plan(rows, DEV)
    assert plan.shuffle_ok
    t = kmer.KmerTable.mini_with_slots(21, DEV, 20, 10).count(s, rows=plan, emit=(1, 8))
    assert t._mini_plan[2] >= len(start) - 200                      # (about) one record per row
    otab = oracle.Table(21, threads=2).count(text)
    assert _same_items(t.items(), otab.items())
    _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=1, vsize=8)
    h = kmer.count_kmers(s, 21, kind="hash")
    _, want = kmer.features(s, rows, k_tnf=None, table=h, window=1, vsize=8, seg_chars=32)
    assert torch.equal(abd, want) and int(abd.sum()) > 30_000


def test_mini_rows_that_cut_reads_and_lowercase():
    """arbitrary row ranges (starting and ending inside reads, with gaps) and soft-masked bases: k-mers that are only valid under
    jellyfish's rule are counted but belong to no row"""
    rng = np.random.RandomState(11)
    runs = []
    for b in range(9):
        seq = bytearray(rng.choice(list(b"ACGT"), size=rng.randint(2500, 6000)).astype(np.uint8).tobytes())
        for _ in range(rng.randint(0, 6)):
            a = rng.randint(0, len(seq)); e = min(len(seq), a + rng.randint(1, 80))
            seq[a:e] = bytes(seq[a:e]).lower()
        for _ in range(3):
            seq[rng.randint(0, len(seq))] = ord("N")
        runs.append((f"bc{b}", bytes(seq) + b"N"))
    s = ReadStream.from_runs(runs, device=DEV)
    text = b"".join(t for _, t in runs)
    lenient = oracle.Table(21, threads=2).count(text.upper())
    cuts = np.sort(rng.choice(np.arange(1, len(text) - 1), size=60, replace=False))
    start, end = cuts[0::2].astype(np.int64), cuts[1::2].astype(np.int64)
    rows = Rows(np.zeros(len(start), dtype=np.int64), [f"r{i}" for i in range(len(start))], start, end)
    plan = kmer.Plan(rows, DEV)
    t = kmer.KmerTable.mini_with_slots(21, DEV, 19, 12).count(s, rows=plan, emit=(1, 64), lowercase_is_base=True)
    assert _same_items(t.items(), lenient.items())
    _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=1, vsize=64)
    _, want = kmer.features(s, rows, k_tnf=None, table=t, window=1, vsize=64, seg_chars=32)       # the lookup kernel: strict by construction
    assert torch.equal(abd, want) and int(abd.sum()) > 0
    strict = kmer.KmerTable.mini_with_slots(21, DEV, 19, 12).count(s, rows=plan, emit=(1, 64))
    assert _same_items(strict.items(), oracle.Table(21, threads=2).count(text).items())


def test_mini_full_bucket_is_reported_and_count_kmers_regrows():
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=10, n_genomes=3, genome_len=60_000, fragment=8_000, seed=9)
    s = synth.generate(cfg, device=DEV)
    small = kmer.KmerTable.mini_with_slots(21, DEV, 12, 8)          # 4096 slots for ~200 k distinct 21-mers
    with pytest.raises(_lib.PangaeaError):
        small.count(s)
    t = kmer.count_kmers(s, 21, kind="mini", distinct_hint=2000)
    assert _same_items(t.items(), oracle.Table(21, threads=4).count(s.decode()).items())


def test_mini_mid_scale_against_oracle():
    """200 k pairs at the bench's geometry (2^15 buckets of 2^14 slots): the whole table and a spread of rows"""
    cfg = synth.SynthConfig(n_pairs=200_000, n_barcodes=1000, n_genomes=8, genome_len=400_000, fragment=40_000, sub_rate=0.002, n_rate=0.01, seed=71)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    t = kmer.KmerTable.mini_with_slots(21, DEV, 29, 14).count(s, rows=plan, emit=(10, 400))
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=10, vsize=400)
    text = s.decode()
    otab = oracle.Table(21, threads=8).count(text)
    assert _same_items(t.items(), otab.items())
    for i in range(0, len(rows), max(1, len(rows) // 25)):
        seq = text[int(rows.start[i]):int(rows.end[i])]
        assert np.array_equal(tnf[i].cpu().numpy(), oracle.tnf_row(seq, 4))
        assert np.array_equal(abd[i].cpu().numpy(), oracle.abd_row(seq, 21, otab, 10, 400))


@pytest.mark.parametrize("k", [21, 25])
def test_mini_low_complexity_runs_and_saturating_counts(k):
    """homopolymers and short tandem repeats: one minimizer value recurs along the whole run (records cut at the cap, one hot
    bucket whose workgroup gets almost every record), 3.1 M copies of one k-mer (the packed count stops at 2^21 and every bin
    below vsize * window stays exact; the wide layout counts on), rows through the emitted words and by lookups"""
    rng = np.random.RandomState(11)
    rnd = bytes(rng.choice(list(b"ACGT"), size=30_000).astype(np.uint8))
    s = ReadStream.from_runs([("a", b"A" * 1_600_000 + b"N" + b"T" * 1_500_040 + b"N"),
                              ("b", b"AC" * 40_000 + b"N" + b"ACG" * 30_000 + b"N" + b"AACCGGTT" * 9_000 + b"N"),
                              ("c", rnd + b"N" + rnd[:20_000] + b"N")], device=DEV)
    rows = s.rows(0)
    plan = kmer.Plan(rows, DEV)
    t = kmer.count_kmers(s, k, rows=plan, emit=(10, 400))
    assert t.kind == ("mini" if k <= 21 else "miniw")
    codes, counts = t.items()
    otab = oracle.Table(k, threads=4).count(s.decode())
    ocodes, ocounts = otab.items()
    assert np.array_equal(codes, ocodes)
    if k <= 21:
        assert counts.max() >= _lib.HASH_COUNT_SAT and np.array_equal(np.minimum(counts, _lib.HASH_COUNT_SAT), np.minimum(ocounts, _lib.HASH_COUNT_SAT))
        assert np.array_equal(counts[ocounts < _lib.HASH_COUNT_SAT], ocounts[ocounts < _lib.HASH_COUNT_SAT])
    else:
        assert np.array_equal(counts, ocounts) and counts.max() > 3_000_000
    _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)
    _, _, oabd = _oracle(s, rows, k, 10, 400)
    assert np.array_equal(abd.cpu().numpy(), oabd)
    _, abd_l = kmer.features(s, rows, k_tnf=None, table=t, window=10, vsize=400)
    assert torch.equal(abd_l, abd)


def test_mini_plan_computed_ahead_on_a_side_stream():
    """``prefetch_plan``: the partition plan of the next count, computed on a side stream (what ``bench.py --plan ahead`` does for
    every batch), is picked up by that count and gives the same table and rows as a plan computed in front of the count"""
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=37, n_genomes=3, genome_len=30_000, fragment=8_000, sub_rate=0.01, n_rate=0.2, seed=808)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    t = kmer.KmerTable.mini_with_slots(21, DEV, 22, 10).count(s, rows=plan, emit=(10, 400))
    want_items = t.items()
    _, want = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)
    first_ws = t._mini_plan[1]
    side = torch.cuda.Stream(device=DEV)
    for _ in range(3):                                               # two workspaces take turns
        t.prefetch_plan(s, plan, side)
        assert t._mini_next is not None and t._mini_next[1] is not t._mini_plan[1]
        t.reset()
        t._mini_plan = None                                          # a new batch: the kept plan does not apply
        t.count(s, rows=plan, emit=(10, 400))
        assert t._mini_next is None
        _, got = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)
        assert torch.equal(got, want) and _same_items(t.items(), want_items)
    assert t._mini_plan[1] is not None and first_ws is not None
    # a prefetched plan for other rows is ignored by a count that does not match it
    other = kmer.Plan(s.rows(0), DEV)
    t.prefetch_plan(s, other, side)
    t.reset().count(s, rows=plan, emit=(10, 400))
    _, got = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)
    assert torch.equal(got, want)


def test_mini_more_rows_than_a_slot_word_holds():
    """300 k rows and buckets of 2^14 slots: row and slot index no longer fit one 32-bit word, so the lookups take the general form
    by themselves (records probed a second time, two-pass row shuffle); rows against the lookup kernel on a key-partitioned table"""
    rng = np.random.RandomState(6)
    n = 300_000
    text = bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
    s = ReadStream.from_runs([("x", text)], device=DEV)
    start = np.arange(50, n - 50, dtype=np.int64)
    rows = Rows(np.zeros(len(start), dtype=np.int64), [""] * len(start), start, start + 1)
    plan = kmer.Plan(rows, DEV)
    assert plan.shuffle_ok and len(start) >= (1 << 18)
    t = kmer.KmerTable.mini_with_slots(21, DEV, 20, 14).count(s, rows=plan, emit=(1, 8))
    _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=1, vsize=8)
    h = kmer.count_kmers(s, 21, kind="hash")
    _, want = kmer.features(s, rows, k_tnf=None, table=h, window=1, vsize=8, seg_chars=32)
    assert torch.equal(abd, want) and int(abd.sum()) > 250_000
    assert _same_items(t.items(), h.items())


@pytest.mark.parametrize("n_rows", [20_000, 40_000, 100_000, 131_072])
def test_one_pass_row_shuffle_for_every_row_count_it_reaches(n_rows, monkeypatch):
    """2^14 < rows <= 2^16 (257 .. 1024 row groups): the (row, bin) words reach their row groups in ONE pass -- in the count kernel
    of the super-k-mer pipeline and in the key-partitioned pipeline's row scatter alike; up to 2^17 rows (2048 groups: 2048 digits in
    the count kernel's lookup tiles) in the super-k-mer pipeline.  PG_S2_TWO_PASS=1 is the two-pass form.  All against the lookup
    kernel; k = 21 and a k > 21 table."""
    rng = np.random.RandomState(n_rows)
    n = 4 * n_rows + 200
    text = bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
    s = ReadStream.from_runs([("x", text)], device=DEV)
    start = 50 + 4 * np.arange(n_rows, dtype=np.int64)
    rows = Rows(np.zeros(n_rows, dtype=np.int64), [""] * n_rows, start, start + 4)
    plan = kmer.Plan(rows, DEV)
    h = kmer.count_kmers(s, 21, kind="hash")
    _, want = kmer.features(s, rows, k_tnf=None, table=h, window=1, vsize=8, seg_chars=32)
    assert int(want.sum()) == n_rows * 4
    for two_pass in (False, True):
        if two_pass:
            monkeypatch.setenv("PG_S2_TWO_PASS", "1")
        t = kmer.KmerTable.mini_with_slots(21, DEV, 20, 12 if n_rows < 50_000 else 14).count(s, rows=plan, emit=(1, 8))
        _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=1, vsize=8)
        assert torch.equal(abd, want)
        b = kmer.KmerTable.with_slots(21, DEV, 20).count(s, rows=plan, emit=(1, 8))
        _, abd_b = kmer.features(s, plan, k_tnf=None, table=b, window=1, vsize=8)
        assert torch.equal(abd_b, want)
    w = kmer.KmerTable.mini_with_slots(25, DEV, 20, 13).count(s, rows=plan, emit=(1, 8))      # (two-pass form: the switch is still set)
    _, abd_w = kmer.features(s, plan, k_tnf=None, table=w, window=1, vsize=8)
    monkeypatch.delenv("PG_S2_TWO_PASS")
    w1 = kmer.KmerTable.mini_with_slots(25, DEV, 20, 13).count(s, rows=plan, emit=(1, 8))
    _, abd_w1 = kmer.features(s, plan, k_tnf=None, table=w1, window=1, vsize=8)
    h25 = kmer.count_kmers(s, 25, kind="wide")
    _, want25 = kmer.features(s, rows, k_tnf=None, table=h25, window=1, vsize=8, seg_chars=32)
    assert torch.equal(abd_w, want25) and torch.equal(abd_w1, want25)


@pytest.mark.parametrize("force", ["5,7", "40,3", "1000,2"])
def test_row_histograms_shared_among_workgroups_give_the_same_rows(force, monkeypatch):
    """the row groups of a last, mostly empty round of workgroups are histogrammed by several workgroups each, which add their
    partial rows (PG_ROW_HIST_SPLIT forces it at this size): same matrix as one workgroup per group -- 2-byte words (one pass),
    4-byte words (two passes), both pipelines"""
    cfg = synth.SynthConfig(n_pairs=30_000, n_barcodes=2500, n_genomes=4, genome_len=50_000, fragment=10_000, sub_rate=0.01, n_rate=0.05, seed=91)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(600)
    plan = kmer.Plan(rows, DEV)
    assert len(rows) > 64 * 30
    got = {}
    for mode in ("0", force):
        monkeypatch.setenv("PG_ROW_HIST_SPLIT", mode)
        for two_pass in (False, True):
            if two_pass:
                monkeypatch.setenv("PG_S2_TWO_PASS", "1")
            else:
                monkeypatch.delenv("PG_S2_TWO_PASS", raising=False)
            t = kmer.KmerTable.mini_with_slots(21, DEV, 22, 12).count(s, rows=plan, emit=(10, 400))
            _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)
            b = kmer.KmerTable.with_slots(21, DEV, 22).count(s, rows=plan, emit=(10, 400))
            _, abd_b = kmer.features(s, plan, k_tnf=None, table=b, window=10, vsize=400)
            got[(mode, two_pass)] = (abd, abd_b)
    ref = got[("0", False)][0]
    assert int(ref.sum()) > 1_000_000
    for v in got.values():
        assert torch.equal(v[0], ref) and torch.equal(v[1], ref)


# ------------------------------------------------------------------ merged (row, bin) runs, the checked build, stale plans


@pytest.mark.parametrize("sub_rate,coverage_genomes,window", [(0.001, 2, 10), (0.05, 2, 1), (0.05, 40, 1)])
def test_merged_runs_equal_the_word_wise_lookups(sub_rate, coverage_genomes, window, monkeypatch):
    """the lookup phase lets runs of equal neighbouring bins of a record travel as one word with a count (MERGE form).  High
    coverage and few errors: nearly every record is one run; window 1 with 5 % substitutions and uneven coverage: neighbours
    rarely share a bin, a tile's runs overflow the LDS stage and are laid out in several windows.  Same rows as the word-wise
    form (PG_MINI_MERGE=0), as the lookup kernel, and as the oracle; both workgroup sizes."""
    cfg = synth.SynthConfig(n_pairs=120_000, n_barcodes=900, n_genomes=coverage_genomes, genome_len=60_000, fragment=20_000,
                            sub_rate=sub_rate, n_rate=0.05, seed=1234)
    s = synth.generate(cfg, device=DEV)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    got = {}
    for lb in (14, 13):
        for merged in (True, False):
            monkeypatch.setenv("PG_MINI_MERGE", "1" if merged else "0")
            t = kmer.KmerTable.mini_with_slots(21, DEV, 26, lb).count(s, rows=plan, emit=(window, 400))
            _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=window, vsize=400)
            got[(lb, merged)] = abd
    monkeypatch.delenv("PG_MINI_MERGE", raising=False)
    ref = got[(14, True)]
    assert all(torch.equal(v, ref) for v in got.values()) and int(ref.sum()) > 10_000_000
    _, want = kmer.features(s, rows, k_tnf=None, table=t, window=window, vsize=400)
    assert torch.equal(ref, want)
    text = s.decode()
    otab = oracle.Table(21, threads=4).count(text)
    for r in range(0, len(rows), max(1, len(rows) // 10)):
        assert np.array_equal(ref[r].cpu().numpy(), oracle.abd_row(text[rows.start[r]:rows.end[r]], 21, otab, window, 400))


@pytest.mark.parametrize("k,log2_slots,lb,piece_words,n_pairs,saturate", [
    (21, 24, 12, 300 * 256, 40_000, False),      # five pieces, the last one short; 4096 buckets
    (21, 26, 14, 1 << 20, 200_000, False),       # two pieces of 2^14-slot buckets (the bench's geometry, 1024-thread workgroups)
    (15, 23, 10, 128 * 256, 15_000, False),      # Pangaea's default k, 11-mer minimizers
    (21, 24, 13, 64 * 256, 0, True),             # counts that reach the saturation value across pieces, a tandem repeat
])
def test_a_stream_counted_in_pieces_equals_one_count(k, log2_slots, lb, piece_words, n_pairs, saturate, monkeypatch):
    """a stream whose scratch does not fit in one piece is counted word range by word range INTO the same table (the bucket's slice
    goes back into LDS, slots keep their places), every piece leaving its provisional slots behind; the lookups of all pieces run when
    the table is final (pg_mini_count_piece / pg_mini_lookup_piece; PANGAEA_MINI_PIECE_WORDS forces the piece size).  Same table and
    same rows as ONE count of the stream, and as the oracle; first through the checked build (test below)."""
    import os
    if os.environ.get("PG_MINI_MERGE", "1") in ("", "0"):
        pytest.skip("the pieces keep the merged lookups' 2-byte slots: no word-wise form")
    if saturate:
        rng = np.random.RandomState(5)
        rnd = bytes(rng.choice(list(b"ACGT"), size=90_000).astype(np.uint8))
        s = ReadStream.from_runs([("a", b"A" * 2_200_000 + b"N" + b"T" * 1_100_040 + b"N"), ("b", b"ACG" * 40_000 + b"N"), ("c", rnd + b"N")], device=DEV)
        rows = s.rows(0)
    else:
        cfg = synth.SynthConfig(n_pairs=n_pairs, n_barcodes=max(8, n_pairs // 150), n_genomes=3, genome_len=80_000, fragment=20_000,
                                sub_rate=0.01, n_rate=0.1, seed=77 + k)
        s = synth.generate(cfg, device=DEV)
        rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    one = kmer.KmerTable.mini_with_slots(k, DEV, log2_slots, lb).count(s, rows=plan, emit=(10, 400))
    assert one._mini_pieces == 1
    _, want = kmer.features(s, plan, k_tnf=None, table=one, window=10, vsize=400)
    monkeypatch.setenv("PANGAEA_MINI_PIECE_WORDS", str(piece_words))
    t = kmer.KmerTable.mini_with_slots(k, DEV, log2_slots, lb)
    t.data.fill_(0x7FFF_FFFF_FFFF)                       # (a fresh table is never cleared: the first piece must not read it)
    t.count(s, rows=plan, emit=(10, 400))
    assert t._mini_pieces == -(-s.n_words // piece_words) >= 2 and t.kind == "mini"
    _, abd = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)            # from the shuffled words of all pieces
    assert torch.equal(abd, want) and int(abd.sum()) > 0
    assert _same_items(t.items(), one.items())
    _, abd_l = kmer.features(s, rows, k_tnf=None, table=t, window=10, vsize=400)          # by lookups in the table the pieces built
    assert torch.equal(abd_l, want)
    text = s.decode()
    otab = oracle.Table(k, threads=4).count(text)
    oc, on = otab.items()
    assert _same_items(t.items(), (oc, np.minimum(on, 1 << 21)))
    if saturate:
        assert int(t.items()[1].max()) == 1 << 21
    for r in range(0, len(rows), max(1, len(rows) // 6)):
        assert np.array_equal(abd[r].cpu().numpy(), oracle.abd_row(text[rows.start[r]:rows.end[r]], k, otab, 10, 400))
    t.reset().count(s, rows=plan, emit=(10, 400))        # again with the same object: the same rows
    _, abd2 = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)
    assert torch.equal(abd2, want)


def test_super_kmer_kernels_through_the_checked_build():
    """the same library built with -DPG_CHECKED (every global store of the super-k-mer kernels checks its index against the
    capacity of the buffer it writes into; PG_STATUS_BOUNDS instead of a memory fault): the oracle cases of this file run
    through it in a process of their own.  New kernel code is run this way first -- see DESIGN.md section 4, 'the abort'."""
    import os
    import subprocess
    import sys
    from .conftest import ROOT
    if os.environ.get("PANGAEA_LIB") == "checked":
        pytest.skip("already inside the checked pass")
    for merge in ("1", "0"):                   # the merged lookups (the default) and the word-wise form
        _checked_pass(merge)


def _checked_pass(merge):
    import os
    import subprocess
    import sys
    from .conftest import ROOT
    env = dict(os.environ, PANGAEA_LIB="checked", PG_MINI_MERGE=merge)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_mini_gpu.py"),
                        "-k", "against_oracle or merged_runs or general_lookup or its_own_record or one_pass_row_shuffle or low_complexity or counted_in_pieces"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_a_plan_of_another_stream_is_not_reused_and_a_forced_one_is_refused(monkeypatch):
    """the cached partition plan is tied to the stream's tensors (their addresses AND versions; the cache holds the tensors, so
    the addresses cannot be recycled) -- a stream rewritten in place gets a new plan; a plan forced onto a stream with more
    records than the record workspace holds sets PG_STATUS_PLAN_MISMATCH and counts nothing"""
    a = synth.generate(synth.SynthConfig(n_pairs=2000, n_barcodes=20, n_genomes=2, genome_len=30_000, fragment=8_000, seed=1), device=DEV)
    b = synth.generate(synth.SynthConfig(n_pairs=2000, n_barcodes=20, n_genomes=2, genome_len=30_000, fragment=8_000, seed=2), device=DEV)
    assert a.n_words == b.n_words
    t = kmer.KmerTable.mini_with_slots(21, DEV, 20, 10).count(a)
    plan_a = t._mini_plan[1]
    a.codes.copy_(b.codes); a.valid.copy_(b.valid)                       # same addresses, other reads
    t.reset().count(a)
    assert t._mini_plan[1] is not plan_a or t._mini_plan[0][1] != 0       # planned again (the version moved on)
    assert _same_items(t.items(), oracle.Table(21, threads=2).count(b.decode()).items())
    # a plan of a small stream forced onto a large one: refused by the kernels
    small = synth.generate(synth.SynthConfig(n_pairs=256, n_barcodes=4, n_genomes=1, genome_len=30_000, fragment=8_000, seed=3), device=DEV)
    pad = torch.zeros(a.n_words - small.n_words, dtype=small.codes.dtype, device=DEV)
    padv = torch.zeros(a.n_words - small.n_words, dtype=small.valid.dtype, device=DEV)
    small_padded = ReadStream(torch.cat([small.codes, pad]), torch.cat([small.valid, padv]), small.n_chars, small.run_off, small.run_names)
    u = kmer.KmerTable.mini_with_slots(21, DEV, 20, 10).count(small_padded)
    monkeypatch.setattr(kmer.KmerTable, "_plan_key", staticmethod(lambda *a_, **k_: "always the same"))
    u.reset()
    u._mini_plan = ("always the same",) + tuple(u._mini_plan[1:])
    with pytest.raises(RuntimeError, match="PLAN_MISMATCH"):
        u.count(a)
