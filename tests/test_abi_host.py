"""CPU-only checks of the C-ABI library: it loads, exports what include/pangaea_feat.h declares, and its host
half (ingest, packing, planning, TNF columns, CSV writer) agrees with the oracle.  No kernel is launched here."""
import ctypes as C
import gzip
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import oracle
from pangaea_amd import _lib, kmer
from pangaea_amd.reads import ReadStream, words_for

from .conftest import GOLDEN, ROOT

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _MAN = json.load(_f)
_INPUTS = {json.dumps(c["input"], sort_keys=True): c["input"] for c in _MAN["cases"]}


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pangaea_feat.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, f"declared in pangaea_feat.h but not exported: {missing}"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert _lib.load().pg_abi_version() == _lib.ABI_VERSION == 9


def test_in_tree_library_is_the_product_build():
    """a variant (make variant ..., possibly wrong results) or a stamped build can never pose as libpangaea_feat.so: its flags say so,
    and load() refuses any library with flags it was not asked for"""
    import subprocess
    import sys
    lib = C.CDLL(os.path.join(ROOT, "pangaea_amd", "libpangaea_feat.so"))
    lib.pg_build_flags.restype = C.c_uint32
    assert lib.pg_build_flags() == 0
    chk = os.path.join(ROOT, "pangaea_amd", "libpangaea_feat_checked.so")
    if os.path.exists(chk):
        code = "from pangaea_amd import _lib; _lib.load()"
        env = dict(os.environ, PANGAEA_LIB=chk)                 # the checked library by PATH, not by name: refused
        env.pop("PANGAEA_ALLOW_VARIANT", None)
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
        assert r.returncode != 0 and "pg_build_flags" in r.stderr
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, PANGAEA_ALLOW_VARIANT="1"), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "pangaea_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn)).read()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, fn


def _np_pack(text: bytes):
    n = len(text)
    nw = words_for(n)
    a = np.frombuffer(text, dtype=np.uint8)
    ok = np.isin(a, np.frombuffer(b"ACGT", dtype=np.uint8))
    base = np.isin(a, np.frombuffer(b"ACGTacgt", dtype=np.uint8))     # lower-case bases keep their code, but are not valid
    code = np.where(base, (a >> 1) & 3, 0).astype(np.uint64)
    pad = nw * 32 - n
    code = np.concatenate([code, np.zeros(pad, np.uint64)]).reshape(nw, 32)
    okp = np.concatenate([ok, np.zeros(pad, bool)]).reshape(nw, 32).astype(np.uint64)
    sh = np.arange(32, dtype=np.uint64)
    return (code << (2 * sh)).sum(axis=1).astype(np.uint64), (okp << sh).sum(axis=1).astype(np.uint32)


def test_pack_ascii_layout():
    rng = np.random.RandomState(1)
    for n in (0, 1, 31, 32, 33, 1000, 8191, 8192, 8193):
        text = bytes(rng.choice(list(b"ACGTNacgtRY\r"), size=n).astype(np.uint8))
        s = ReadStream.from_runs([("x", text)])
        cw, vw = _np_pack(text)
        assert s.n_words % _lib.WORD_ALIGN == 0 and s.n_words >= 1
        assert np.array_equal(s.codes.numpy().view(np.uint64), cw)
        assert np.array_equal(s.valid.numpy().view(np.uint32), vw)
        want = bytes(c if c in b"ACGT" else ord("N") for c in text)
        assert s.decode() == want


@pytest.mark.parametrize("spec", list(_INPUTS.values()), ids=lambda s: s.get("i") or s["1"])
def test_ingest_matches_oracle_runs(spec):
    r1 = os.path.join(GOLDEN, spec.get("i") or spec["1"])
    r2 = os.path.join(GOLDEN, spec["2"]) if "2" in spec else None
    rd = oracle.Reads(r1, r2)
    s = ReadStream.from_fastq(r1, r2)
    assert s.run_names == rd.names
    assert np.array_equal(s.run_off, rd.seq_off)
    assert (s.n_pairs, s.n_unpaired, s.mode) == (rd.n_pairs, rd.n_unpaired, rd.mode)
    norm = lambda b: bytes(c if c in b"ACGT" else ord("N") for c in b)
    for i in range(rd.n_runs):
        assert s.decode(int(s.run_off[i]), int(s.run_off[i + 1])) == norm(rd.seq(i))
    # every read of the input is in the stream exactly once (what the global counter sees)
    k = 9
    a = oracle.Table(k).count(rd.all_seq())
    # (paired files: jellyfish's view leaves out the bases below its quality threshold -- the stream's quality plane)
    b = oracle.Table(k).count(s.decode(plane=s.table_valid(lowercase_is_base=False)))
    assert all(np.array_equal(x, y) for x, y in zip(a.items(), b.items()))
    assert (s.valid_lowq is not None) == ("pairq" in r1)
    for mlen in (0, 100, 168, 2000):
        assert list(s.rows(mlen).run_index) == rd.surviving(mlen)


def test_ingest_errors_are_reported_not_fatal(tmp_path):
    L = _lib.load()
    h = C.c_void_p()
    assert L.pg_ingest_fastq(b"/nonexistent/file.fq", None, C.byref(h)) == -2
    assert b"cannot open" in L.pg_last_error()
    bad = tmp_path / "bad.fq"
    bad.write_text("@r1 BX:Z\nACGT\n+\nIIII\n@r1 BX:Z\nACGT\n+\nIIII\n")
    assert L.pg_ingest_fastq(str(bad).encode(), None, C.byref(h)) == -3
    with pytest.raises(_lib.PangaeaError):
        ReadStream.from_fastq(str(bad))
    with pytest.raises(RuntimeError):           # oracle refuses the same input
        oracle.Reads(str(bad))


def test_plan_segments_partitions_rows():
    rng = np.random.RandomState(3)
    start = np.cumsum(rng.randint(1, 5000, size=50)).astype(np.int64)
    end = start + rng.randint(0, 4000, size=50)
    from pangaea_amd.reads import Rows
    rows = Rows(np.arange(50), [str(i) for i in range(50)], start, end)
    for seg in (32, 64, 1024):
        r, s, e = kmer.plan_segments(rows, seg)
        assert ((e - s) > 0).all() and ((e - s) <= seg).all()
        for i in range(50):
            m = r == i
            if end[i] == start[i]:
                assert not m.any()
                continue
            assert s[m][0] == start[i] and e[m][-1] == end[i] and np.array_equal(s[m][1:], e[m][:-1])
    L = _lib.load()
    assert L.pg_plan_segments(start.ctypes.data, end.ctypes.data, 50, 48, None, None, None) == -1


def test_tnf_columns_follow_reference_order():
    for k in range(1, _lib.TNF_MAX_K + 1):
        colmap, codes = kmer.tnf_colmap(k)
        assert np.array_equal(codes, oracle.tnf_columns(k))
        cm = colmap.numpy().view(np.uint16)
        for c in range(4 ** k):
            canon = min(c, oracle.revcomp(c, k))
            assert codes[cm[c]] == canon
    assert kmer.tnf_ncols(4) == 136
    with pytest.raises(_lib.PangaeaError):
        kmer.tnf_ncols(7)


def test_csv_writer_prints_like_the_reference(tmp_path):
    # golden rows of the poly-A input hold a count above 999999 (exponent form) -- rebuild them from integers
    with open(os.path.join(GOLDEN, "polya.tnf.k4.l1000.csv")) as f:
        golden = f.read()
    rd = oracle.Reads(os.path.join(GOLDEN, "polya.fq.gz"))
    names, tnf, _ = rd.features(1000, k_tnf=4)
    mat = np.ascontiguousarray(tnf, dtype=np.int32)
    blob = b"".join(n.encode() + b"\0" for n in names)
    out = str(tmp_path / "t.gz")
    _lib.check(_lib.load().pg_write_csv_gz(out.encode(), blob, mat.ctypes.data, mat.shape[0], mat.shape[1]))
    with gzip.open(out, "rt") as f:
        assert f.read() == golden
    assert "1.23981e+06" in golden


def test_kernels_refuse_cpu_tensors():
    s = ReadStream.from_runs([("x", b"ACGTACGTACGTN")])
    with pytest.raises(RuntimeError, match="no CPU path"):
        kmer.features(s, s.rows(0), k_tnf=4)
    assert [kmer.KmerTable.default_kind(k) for k in (4, 8, 9, 15, 21, 22, 31)] == ["dense", "dense", "hash", "hash", "hash", "wide", "wide"]
    with pytest.raises(ValueError):
        kmer.KmerTable.default_kind(32)


def test_parallel_ingest_equals_serial(tmp_path):
    """the threaded interleaved parser (line index, latch scan, parse, bit-offset packing, run stitching) gives the very
    same stream, runs and counters as the sequential loop, for any thread count, also when the grammar latches late"""
    from pangaea_amd import synth
    L = _lib.load()
    cfg = synth.SynthConfig(n_pairs=9000, n_barcodes=41, n_genomes=2, genome_len=30_000, fragment=5_000, n_rate=0.3, unbarcoded=0.05)
    fq = str(tmp_path / "a.fq")
    synth.write_fastq(synth.generate(cfg), cfg, fq)
    # prepend untagged pairs (grammar undecided for a while) and cut the last record short
    text = open(fq).read()
    head = "".join(f"@u{i}/1\nACGTNACGT\n+\nIIIIIIIII\n@u{i}/2\nTTGCA\n+\nIIIII\n" for i in range(700))
    lines = (head + text).splitlines(keepends=True)
    open(fq, "w").write("".join(lines[:-5]))
    try:
        L.pg_set_ingest_threads(1)
        ref = ReadStream.from_fastq(fq)
        for T in (2, 3, 7, 16):
            L.pg_set_ingest_threads(T)
            got = ReadStream.from_fastq(fq)
            assert np.array_equal(ref.codes.numpy(), got.codes.numpy()) and np.array_equal(ref.valid.numpy(), got.valid.numpy())
            assert np.array_equal(ref.run_off, got.run_off) and ref.run_names == got.run_names
            assert (ref.n_chars, ref.n_pairs, ref.mode) == (got.n_chars, got.n_pairs, got.mode)
        # gzip input: inflated once (into an in-memory file), then parsed by the same threads
        gz = str(tmp_path / "a.fq.gz")
        with gzip.open(gz, "wb", compresslevel=1) as f:
            f.write(open(fq, "rb").read())
        for T in (1, 4):
            L.pg_set_ingest_threads(T)
            got = ReadStream.from_fastq(gz)
            assert np.array_equal(ref.codes.numpy(), got.codes.numpy()) and np.array_equal(ref.valid.numpy(), got.valid.numpy())
            assert np.array_equal(ref.run_off, got.run_off) and ref.run_names == got.run_names and ref.n_pairs == got.n_pairs
        # a last line without its newline (the threaded reader used to hand out stale line positions there)
        bare = str(tmp_path / "bare.fq")
        open(bare, "w").write(open(fq).read().rstrip("\n"))
        L.pg_set_ingest_threads(1)
        want = ReadStream.from_fastq(bare)
        for T in (2, 5):
            L.pg_set_ingest_threads(T)
            got = ReadStream.from_fastq(bare)
            assert np.array_equal(want.codes.numpy(), got.codes.numpy()) and np.array_equal(want.valid.numpy(), got.valid.numpy())
            assert np.array_equal(want.run_off, got.run_off) and want.run_names == got.run_names and want.n_pairs == got.n_pairs
    finally:
        L.pg_set_ingest_threads(0)
    rd = oracle.Reads(fq)
    assert ref.run_names == rd.names and np.array_equal(ref.run_off, rd.seq_off) and ref.n_pairs == rd.n_pairs


def _shard_union(path, n_parts):
    """(names, run texts, pairs, modes) of the shards of ``path`` in rank order"""
    names, texts, pairs, modes = [], [], 0, set()
    for r in range(n_parts):
        s = ReadStream.from_fastq_shard(path, r, n_parts)
        assert int(s.run_off[-1]) == s.n_chars          # interleaved: nothing outside the runs
        names += s.run_names
        texts += [s.decode(int(s.run_off[i]), int(s.run_off[i + 1])) for i in range(len(s.run_names))]
        pairs += s.n_pairs
        modes.add(s.mode)
    return names, texts, pairs, modes


def _whole(path):
    s = ReadStream.from_fastq(path)
    texts = [s.decode(int(s.run_off[i]), int(s.run_off[i + 1])) for i in range(len(s.run_names))]
    return s, texts


@pytest.mark.parametrize("spec", [s for s in _INPUTS.values() if "i" in s], ids=lambda s: s["i"])
def test_sharded_ingest_reassembles_the_whole_file(spec, tmp_path):
    """byte-range shards (any count, boundaries anywhere: inside headers, quality lines that start with '@', runs that
    span several shards, the unbarcoded tail, a grammar that latches late) concatenate to the whole-file ingest"""
    src = os.path.join(GOLDEN, spec["i"])
    path = str(tmp_path / "plain.fq")
    with (gzip.open(src, "rb") if src.endswith(".gz") else open(src, "rb")) as f:
        open(path, "wb").write(f.read())
    whole, texts = _whole(path)
    for n_parts in (1, 2, 3, 5, 8, 13):
        names, got, pairs, modes = _shard_union(path, n_parts)
        assert names == whole.run_names and got == texts and pairs == whole.n_pairs
        assert modes == {whole.mode}


def test_sharded_ingest_large_file_and_errors(tmp_path):
    from pangaea_amd import synth
    L = _lib.load()
    cfg = synth.SynthConfig(n_pairs=6000, n_barcodes=37, n_genomes=2, genome_len=30_000, fragment=5_000, n_rate=0.2, unbarcoded=0.2)
    path = str(tmp_path / "a.fq")
    synth.write_fastq(synth.generate(cfg), cfg, path)
    # quality lines made of '@' only: a boundary that lands there must not be taken for a header
    lines = open(path).read().splitlines(keepends=True)
    for i in range(3, len(lines), 4):
        lines[i] = "@" * (len(lines[i]) - 1) + "\n"
    head = "".join(f"@u{i}/1\nACGTNACGT\n+\n@@@@@@@@@\n@u{i}/2\nTTGCA\n+\n@@@@@\n" for i in range(300))
    open(path, "w").write(head + "".join(lines[:-5]))          # late latch + a record cut short at the end
    L.pg_set_ingest_threads(1)
    whole, texts = _whole(path)                     # the sequential loop
    try:
        for T, block in ((1, None), (4, None), (3, "16"), (5, "4096")):      # tiny blocks: refills, growth, units across blocks
            L.pg_set_ingest_threads(T)
            if block:
                os.environ["PG_INGEST_BLOCK"] = block
            for n_parts in (2, 7, 8, 64):
                names, got, pairs, _ = _shard_union(path, n_parts)
                assert names == whole.run_names and got == texts and pairs == whole.n_pairs
            if T > 1:
                again = ReadStream.from_fastq(path)     # the threaded whole-file path through the same readers
                assert np.array_equal(again.codes.numpy(), whole.codes.numpy()) and np.array_equal(again.valid.numpy(), whole.valid.numpy())
                assert again.run_names == whole.run_names and np.array_equal(again.run_off, whole.run_off)
    finally:
        L.pg_set_ingest_threads(0)
        os.environ.pop("PG_INGEST_BLOCK", None)
    # the counts a caller exchanges: prefix sums over the ranges are the whole file's newline count
    assert sum(ReadStream.count_newlines(path, r, 5) for r in range(5)) == open(path, "rb").read().count(b"\n")
    # one barcode only: the run in progress never ends, every later shard is empty and rank 0 carries the lot
    one = str(tmp_path / "one.fq")
    open(one, "w").write("".join(f"@r{i} BX:Z:AAAA-1\nACGTACGTAC\n+\nIIIIIIIIII\n@r{i} BX:Z:AAAA-1\nGGGTTTAAAC\n+\nIIIIIIIIII\n" for i in range(50)))
    w1, t1 = _whole(one)
    names, got, pairs, _ = _shard_union(one, 4)
    assert names == w1.run_names and got == t1 and pairs == 50
    # gzip cannot be cut by bytes
    gz = str(tmp_path / "a.fq.gz")
    with gzip.open(gz, "wb") as f:
        f.write(open(one, "rb").read())
    with pytest.raises(RuntimeError, match="uncompressed"):
        ReadStream.from_fastq_shard(gz, 0, 2)
    with pytest.raises(RuntimeError, match="uncompressed"):
        ReadStream.count_newlines(gz, 0, 2)
    with pytest.raises(RuntimeError, match="cannot open"):
        ReadStream.count_newlines(str(tmp_path / "missing.fq"), 0, 2)


def test_packed_stream_cache_round_trip(tmp_path):
    """ReadStream.save / load: the packed stream, its runs and counters come back identical (arrays memory-mapped)"""
    for name, r2 in (("tenx_mixed.fq", None), ("stlfr.fq", None), ("pair_R1.fq", "pair_R2.fq")):
        s = ReadStream.from_fastq(os.path.join(GOLDEN, name), os.path.join(GOLDEN, r2) if r2 else None)
        path = str(tmp_path / (name + ".pgstream"))
        s.save(path)
        t = ReadStream.load(path)
        assert np.array_equal(s.codes.numpy(), t.codes.numpy()) and np.array_equal(s.valid.numpy(), t.valid.numpy())
        assert np.array_equal(s.run_off, t.run_off) and s.run_names == t.run_names
        assert (s.n_chars, s.n_pairs, s.n_unpaired, s.mode) == (t.n_chars, t.n_pairs, t.n_unpaired, t.mode)
        assert t.decode() == s.decode()
    empty = ReadStream.from_runs([])
    empty.save(str(tmp_path / "e.pgstream"))
    assert ReadStream.load(str(tmp_path / "e.pgstream")).n_chars == 0
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 5)
    with pytest.raises(ValueError, match="truncated"):
        ReadStream.load(path)
    open(path, "wb").write(b"@r1\nACGT\n")
    with pytest.raises(ValueError, match="not a packed read stream"):
        ReadStream.load(path)


def test_sharded_ingest_of_degenerate_files(tmp_path):
    """an empty file, a single pair and a file that ends inside its first record: any number of shards gives the whole-file runs"""
    cases = {"empty.fq": "", "one.fq": "@r BX:Z:AAAA-1\nACGT\n+\nIIII\n@r BX:Z:AAAA-1\nTTGG\n+\nIIII\n", "partial.fq": "@r BX:Z:AAAA-1\nACGT\n+\n"}
    for name, content in cases.items():
        path = str(tmp_path / name)
        open(path, "w").write(content)
        whole = ReadStream.from_fastq(path)
        for parts in (1, 2, 5):
            shards = [ReadStream.from_fastq_shard(path, r, parts) for r in range(parts)]
            assert [n for s in shards for n in s.run_names] == whole.run_names
            assert sum(s.n_chars for s in shards) == whole.n_chars and sum(s.n_pairs for s in shards) == whole.n_pairs


def test_lower_case_plane_of_the_ingest(tmp_path):
    """lower-case a c g t: not valid (the reference's counters reset on them), but marked in `valid_lower` with their codes in
    place, identically by the sequential loop, the threaded parser (any block size) and byte-range shards"""
    from pangaea_amd import synth
    L = _lib.load()
    cfg = synth.SynthConfig(n_pairs=5000, n_barcodes=31, n_genomes=2, genome_len=30_000, fragment=5_000, n_rate=0.05, seed=8)
    fq = str(tmp_path / "soft.fq")
    synth.write_fastq(synth.generate(cfg), cfg, fq)
    rng = np.random.RandomState(3)
    lines = open(fq).read().splitlines(keepends=True)
    for i in range(1, len(lines), 4):                             # soft-mask stretches of every third read
        if rng.rand() < 0.33:
            seq = lines[i].rstrip("\n")
            a = rng.randint(0, len(seq)); b = min(len(seq), a + rng.randint(1, 60))
            lines[i] = seq[:a] + seq[a:b].lower() + seq[b:] + "\n"
    open(fq, "w").write("".join(lines))
    try:
        L.pg_set_ingest_threads(1)
        ref = ReadStream.from_fastq(fq)
        assert ref.valid_lower is not None
        text = b"".join((l.rstrip("\n") + "N").encode() for l in lines[1::4])
        want_lower = np.array([c in b"acgt" for c in text])
        sh = np.arange(32, dtype=np.uint32)
        got_lower = ((ref.valid_lower.numpy().view(np.uint32)[:, None] >> sh[None, :]) & 1).astype(bool).ravel()[:len(text)]
        got_valid = ((ref.valid.numpy().view(np.uint32)[:, None] >> sh[None, :]) & 1).astype(bool).ravel()[:len(text)]
        assert np.array_equal(got_lower, want_lower) and not (got_lower & got_valid).any()
        codes = ((ref.codes.numpy().view(np.uint64)[:, None] >> (2 * sh.astype(np.uint64))[None, :]) & np.uint64(3)).ravel()[:len(text)]
        up = np.frombuffer(text.upper(), dtype=np.uint8)
        base = np.isin(up, np.frombuffer(b"ACGT", dtype=np.uint8))
        assert np.array_equal(codes[base], ((up[base] >> 1) & 3).astype(np.uint64))
        assert ref.decode() == bytes(c if c in b"ACGT" else ord("N") for c in text)          # the strict view is unchanged
        for threads, block in ((4, None), (7, "1000")):
            L.pg_set_ingest_threads(threads)
            if block:
                os.environ["PG_INGEST_BLOCK"] = block
            got = ReadStream.from_fastq(fq)
            assert np.array_equal(got.codes.numpy(), ref.codes.numpy()) and np.array_equal(got.valid.numpy(), ref.valid.numpy())
            assert np.array_equal(got.valid_lower.numpy(), ref.valid_lower.numpy())
            parts = [ReadStream.from_fastq_shard(fq, r, 3) for r in range(3)]
            assert sum(int(torch.count_nonzero(p.valid_lower)) > 0 for p in parts if p.valid_lower is not None) >= 1
            tot = sum(sum(bin(int(w) & 0xFFFFFFFF).count("1") for w in p.valid_lower.numpy()) for p in parts if p.valid_lower is not None)
            assert tot == int(want_lower.sum())
        # the cache keeps the plane
        ref.save(str(tmp_path / "s.pgstream"))
        back = ReadStream.load(str(tmp_path / "s.pgstream"))
        assert np.array_equal(back.valid_lower.numpy(), ref.valid_lower.numpy())
    finally:
        L.pg_set_ingest_threads(0)
        os.environ.pop("PG_INGEST_BLOCK", None)
    plain = ReadStream.from_fastq(os.path.join(GOLDEN, "tenx_clean.fq.gz"))
    assert plain.valid_lower is None


def _same_stream(a, b):
    def plane(x):
        return None if x is None else x.numpy()
    ok = np.array_equal(a.codes.numpy(), b.codes.numpy()) and np.array_equal(a.valid.numpy(), b.valid.numpy())
    for x, y in ((a.valid_lower, b.valid_lower), (a.valid_lowq, b.valid_lowq)):
        ok = ok and ((x is None) == (y is None)) and (x is None or np.array_equal(plane(x), plane(y)))
    return (ok and np.array_equal(a.run_off, b.run_off) and a.run_names == b.run_names
            and (a.n_chars, a.n_pairs, a.n_unpaired, a.mode) == (b.n_chars, b.n_pairs, b.n_unpaired, b.mode))


def test_parallel_paired_ingest_equals_serial(tmp_path):
    """-1 / -2 input through the threaded reader (records cut by R1 byte ranges, the same record located in R2, the serial rules
    on the tail) gives the very same stream, planes (lower case, quality below '?'), runs and counters as the sequential loop:
    mismatched names and barcodes, a grammar that latches late and in R2 first, ragged read lengths, a record cut short, an R2
    that is shorter or longer than R1, CRLF"""
    from pangaea_amd import synth
    L = _lib.load()
    rs = np.random.RandomState(12)
    cfg = synth.SynthConfig(n_pairs=6000, n_barcodes=29, n_genomes=2, genome_len=30_000, fragment=5_000, n_rate=0.3, unbarcoded=0.05)
    fq = str(tmp_path / "i.fq")
    synth.write_fastq(synth.generate(cfg), cfg, fq)
    lines = open(fq).read().splitlines()
    recs = [lines[i:i + 4] for i in range(0, len(lines), 4)]
    r1, r2 = recs[0::2], recs[1::2]

    def qual(n):
        return "".join(rs.choice(list("#+5>?@FI"), size=n))

    def soften(s):
        a = rs.randint(0, max(1, len(s) - 5))
        return s[:a] + s[a:a + 4].lower() + s[a + 4:]
    for i, (a, b) in enumerate(zip(r1, r2)):
        a[1] = a[1][:rs.randint(20, len(a[1]) + 1)]
        a[3], b[3] = qual(len(a[1])), qual(len(b[1]) - (i % 7 == 0))          # (some quality lines one short)
        if i % 11 == 0:
            a[1] = soften(a[1])
        if i % 13 == 0:
            b[0] = b[0].replace("@", "@x", 1)                                     # other name: the pair is skipped
        if i % 17 == 0 and "BX:Z:" in b[0]:
            b[0] = b[0].replace("BX:Z:A", "BX:Z:C").replace("BX:Z:G", "BX:Z:T")  # other barcode: skipped
    head1 = [[f"@u{i}", "ACGTNACGTTTGA", "+", "IIIII5IIIIIII"] for i in range(500)]
    head2 = [[f"@u{i}", "TTGCANNA", "+", "II#IIIII"] for i in range(500)]
    head2[320][0] += " BX:Z:ACGT-1"                                              # the grammar latches on an R2 header first

    def text(rr, eol="\n"):
        return "".join(eol.join(r) + eol for r in rr)
    variants = {
        "even": (text(head1 + r1), text(head2 + r2)),
        "cut": (text(head1 + r1)[:-37], text(head2 + r2)),                       # R1's last record cut inside its quality line
        "short2": (text(head1 + r1), text(head2 + r2[:-40] + [r2[-40][:2]])),    # R2 ends early, inside a record
        "long2": (text(head1 + r1[:-25]), text(head2 + r2)),                     # R2 records beyond the end of R1
        "crlf": (text(head1 + r1, "\r\n"), text(head2 + r2, "\r\n")),
        "noeol": (text(head1 + r1).rstrip("\n"), text(head2 + r2).rstrip("\n")),
    }
    try:
        for name, (t1, t2) in variants.items():
            p1, p2 = str(tmp_path / f"{name}_1.fq"), str(tmp_path / f"{name}_2.fq")
            open(p1, "w", newline="").write(t1)
            open(p2, "w", newline="").write(t2)
            L.pg_set_ingest_threads(1)
            ref = ReadStream.from_fastq(p1, p2)
            assert ref.n_unpaired > 100 and ref.valid_lowq is not None and ref.valid_lower is not None, name
            for T, block in ((2, None), (5, "100"), (8, None), (13, "4096")):
                L.pg_set_ingest_threads(T)
                if block:
                    os.environ["PG_INGEST_BLOCK"] = block
                else:
                    os.environ.pop("PG_INGEST_BLOCK", None)
                got = ReadStream.from_fastq(p1, p2)
                assert _same_stream(ref, got), (name, T, block)
            if name in ("even", "long2"):                       # gzip of one file or of both: inflated side by side, then threaded
                for which in ((1,), (1, 2)):
                    q = [p1, p2]
                    for w in which:
                        q[w - 1] = [p1, p2][w - 1] + ".gz"
                        with gzip.open(q[w - 1], "wb", compresslevel=1) as f:
                            f.write(open([p1, p2][w - 1], "rb").read())
                    L.pg_set_ingest_threads(6)
                    assert _same_stream(ref, ReadStream.from_fastq(q[0], q[1])), (name, which)
    finally:
        L.pg_set_ingest_threads(0)
        os.environ.pop("PG_INGEST_BLOCK", None)
    rd = oracle.Reads(p1, p2)
    assert ref.run_names == rd.names and ref.n_pairs == rd.n_pairs


def test_gzip_text_parked_in_memory_is_the_text(tmp_path, monkeypatch):
    """pg_inflate_to_memfd (what the device ingest reads gzip input through): the descriptor holds the inflated bytes of a
    one- and a two-member file; a plain file and a text beyond the budget give no descriptor and no error"""
    import ctypes as C
    import gzip
    L = _lib.load()
    text = b"".join(b"@r%d BX:Z:ACGT-1\nACGTNACGT\n+\nIIIIIIIII\n" % i for i in range(5000))
    cases = {"one.gz": gzip.compress(text), "two.gz": gzip.compress(text[:70000]) + gzip.compress(text[70000:]), "plain.fq": text}
    for name, data in cases.items():
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        fd, size = C.c_int(-7), C.c_int64(-7)
        _lib.check(L.pg_inflate_to_memfd(path.encode(), C.byref(fd), C.byref(size)))
        if name == "plain.fq":
            assert fd.value == -1 and size.value == 0
            continue
        assert fd.value >= 0 and size.value == len(text)
        try:
            assert open(f"/proc/self/fd/{fd.value}", "rb").read() == text
        finally:
            os.close(fd.value)
    monkeypatch.setenv("PG_INFLATE_MAX_BYTES", "4096")
    fd, size = C.c_int(-7), C.c_int64(-7)
    _lib.check(L.pg_inflate_to_memfd(str(tmp_path / "one.gz").encode(), C.byref(fd), C.byref(size)))
    assert fd.value == -1 and size.value == 0
    with pytest.raises(RuntimeError):
        _lib.check(L.pg_inflate_to_memfd(str(tmp_path / "missing.gz").encode(), C.byref(fd), C.byref(size)))
