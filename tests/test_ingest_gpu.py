"""The ingest with the device copy inside (`pg_ingest_fastq_device` + `pg_ingest_place`, ingest_dev.hip): parser threads copy
finished pieces to the GPU while the others parse on, the shift into place is a kernel.  The stream, runs and counters must be
those of the host ingest (which the CPU suite pins against the oracle and the reference's goldens) for every piece size, thread
count and reader block -- pieces of a few hundred bytes make words that collect characters from many pieces, empty pieces
and pieces that end inside a record."""
import ctypes as C
import glob
import gzip
import os

import numpy as np
import pytest
import torch

from pangaea_amd import _lib, synth
from pangaea_amd.reads import ReadStream

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


def _same(dev_stream, host_stream):
    assert dev_stream.codes.is_cuda and dev_stream.valid.is_cuda
    assert (dev_stream.n_chars, dev_stream.n_pairs, dev_stream.n_unpaired, dev_stream.mode) == \
           (host_stream.n_chars, host_stream.n_pairs, host_stream.n_unpaired, host_stream.mode)
    assert dev_stream.run_names == host_stream.run_names and np.array_equal(dev_stream.run_off, host_stream.run_off)
    n = (host_stream.n_chars + 31) // 32
    assert dev_stream.n_words >= n and dev_stream.n_words % _lib.WORD_ALIGN == 0
    got_c, got_v = dev_stream.codes.cpu().numpy(), dev_stream.valid.cpu().numpy()
    assert np.array_equal(got_c[:n], host_stream.codes.numpy()[:n]) and np.array_equal(got_v[:n], host_stream.valid.numpy()[:n])
    assert not got_c[n:].any() and not got_v[n:].any()                  # padding words are zero
    assert (dev_stream.valid_lower is None) == (host_stream.valid_lower is None)
    if host_stream.valid_lower is not None:
        assert np.array_equal(dev_stream.valid_lower.cpu().numpy()[:n], host_stream.valid_lower.numpy()[:n])
    assert (dev_stream.valid_lowq is None) == (host_stream.valid_lowq is None)
    if host_stream.valid_lowq is not None:
        got_q = dev_stream.valid_lowq.cpu().numpy()
        assert np.array_equal(got_q[:n], host_stream.valid_lowq.numpy()[:n]) and not got_q[n:].any()


def _plain_goldens(tmp_path):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.fq*"))):
        if "_R1" in p or "_R2" in p:
            continue
        if p.endswith(".gz"):
            q = str(tmp_path / os.path.basename(p)[:-3])
            open(q, "wb").write(gzip.open(p, "rb").read())
            p = q
        out.append(p)
    return out


@pytest.fixture
def knobs():
    L = _lib.load()
    yield L
    L.pg_set_ingest_threads(0)
    for k in ("PG_INGEST_PIECE", "PG_INGEST_BLOCK", "PANGAEA_INGEST_ON_HOST"):
        os.environ.pop(k, None)


def test_device_ingest_of_the_golden_inputs(tmp_path, knobs):
    L = knobs
    files = _plain_goldens(tmp_path)
    assert len(files) >= 5
    for path in files:
        L.pg_set_ingest_threads(1)
        os.environ["PANGAEA_INGEST_ON_HOST"] = "1"
        try:
            want = ReadStream.from_fastq(path)
        except RuntimeError:
            os.environ.pop("PANGAEA_INGEST_ON_HOST")
            with pytest.raises(RuntimeError):                            # (the input the reference aborts on: same refusal)
                ReadStream.from_fastq(path, device=DEV)
            continue
        os.environ.pop("PANGAEA_INGEST_ON_HOST")
        for threads, piece in ((1, None), (3, "64"), (4, "200"), (2, "1000"), (7, "4096")):
            L.pg_set_ingest_threads(threads)
            if piece:
                os.environ["PG_INGEST_PIECE"] = piece
            else:
                os.environ.pop("PG_INGEST_PIECE", None)
            _same(ReadStream.from_fastq(path, device=DEV), want)


def test_device_ingest_of_a_larger_file_with_a_late_latch_soft_masking_and_a_cut_record(tmp_path, knobs):
    L = knobs
    cfg = synth.SynthConfig(n_pairs=60_000, n_barcodes=211, n_genomes=2, genome_len=50_000, fragment=5_000, n_rate=0.2, unbarcoded=0.05, seed=5)
    fq = str(tmp_path / "a.fq")
    synth.write_fastq(synth.generate(cfg), cfg, fq)
    lines = open(fq).read().splitlines(keepends=True)
    rng = np.random.RandomState(4)
    for i in range(1, len(lines), 4):
        if rng.rand() < 0.1:                                             # soft-masked stretches
            seq = lines[i].rstrip("\n")
            a = rng.randint(0, len(seq)); b = min(len(seq), a + rng.randint(1, 60))
            lines[i] = seq[:a] + seq[a:b].lower() + seq[b:] + "\n"
    for i in range(3, len(lines), 4):
        if rng.rand() < 0.3:
            lines[i] = "@" * (len(lines[i]) - 1) + "\n"                   # quality lines that look like headers
    head = "".join(f"@u{i}/1\nACGTNACGT\n+\nIIIIIIIII\n@u{i}/2\nTTGCA\n+\nIIIII\n" for i in range(700))
    open(fq, "w").write(head + "".join(lines[:-5]))
    L.pg_set_ingest_threads(1)
    os.environ["PANGAEA_INGEST_ON_HOST"] = "1"
    want = ReadStream.from_fastq(fq)
    os.environ.pop("PANGAEA_INGEST_ON_HOST")
    assert want.valid_lower is not None and want.n_pairs > 50_000
    for threads, piece, block in ((1, None, None), (8, None, None), (5, "65536", None), (16, "300000", "4096"), (3, "1000000", "100")):
        L.pg_set_ingest_threads(threads)
        for k, v in (("PG_INGEST_PIECE", piece), ("PG_INGEST_BLOCK", block)):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        _same(ReadStream.from_fastq(fq, device=DEV), want)
        for n_parts in (3, 8):
            before = np.concatenate([[0], np.cumsum([ReadStream.count_newlines(fq, i, n_parts) for i in range(n_parts)])])
            for r in range(n_parts):
                os.environ["PANGAEA_INGEST_ON_HOST"] = "1"
                host = ReadStream.from_fastq_shard(fq, r, n_parts, before)
                os.environ.pop("PANGAEA_INGEST_ON_HOST")
                _same(ReadStream.from_fastq_shard(fq, r, n_parts, before, device=DEV), host)
    # a last line without its newline
    bare = str(tmp_path / "bare.fq")
    open(bare, "w").write(open(fq).read().rstrip("\n"))
    L.pg_set_ingest_threads(1)
    want = ReadStream.from_fastq(bare)
    L.pg_set_ingest_threads(6)
    os.environ["PG_INGEST_PIECE"] = "70000"
    _same(ReadStream.from_fastq(bare, device=DEV), want)


def test_device_ingest_of_degenerate_and_other_inputs(tmp_path, knobs):
    L = knobs
    rec = "@a BX:Z:AAAC-1\nACGTACGTAC\n+\nIIIIIIIIII\n@a BX:Z:AAAC-1\nTTTTGGGGCC\n+\nIIIIIIIIII\n"
    cases = {"empty.fq": "", "one.fq": rec, "cut.fq": rec[:30], "two.fq": rec + rec.replace("AAAC", "CCCA"), "nolf.fq": (rec + rec).rstrip("\n")}
    for name, text in cases.items():
        path = str(tmp_path / name)
        open(path, "w").write(text)
        L.pg_set_ingest_threads(1)
        want = ReadStream.from_fastq(path)
        for threads, piece in ((1, None), (4, "64")):
            L.pg_set_ingest_threads(threads)
            if piece:
                os.environ["PG_INGEST_PIECE"] = piece
            else:
                os.environ.pop("PG_INGEST_PIECE", None)
            _same(ReadStream.from_fastq(path, device=DEV), want)
    os.environ.pop("PG_INGEST_PIECE", None)
    # gzip input: inflated into an in-memory file, then the same pieces and copies (test_gzip_input_... below)
    gz = str(tmp_path / "two.fq.gz")
    with gzip.open(gz, "wb") as f:
        f.write(cases["two.fq"].encode())
    _same(ReadStream.from_fastq(gz, device=DEV), ReadStream.from_fastq(gz))
    # errors come back as errors
    with pytest.raises(RuntimeError):
        ReadStream.from_fastq(str(tmp_path / "missing.fq"), device=DEV)
    bad = str(tmp_path / "bad.fq")
    open(bad, "w").write(rec.replace("BX:Z:AAAC-1\nACGT", "BX:Z\nACGT", 1))
    with pytest.raises(RuntimeError):
        ReadStream.from_fastq(bad, device=DEV)
    # staging arrays that are too small are refused, not overrun
    path = str(tmp_path / "two.fq")
    small = torch.zeros(8, dtype=torch.int64, device=DEV)
    h = C.c_void_p()
    rc = L.pg_ingest_fastq_device(path.encode(), 0, 1, None, os.path.getsize(path), C.c_void_p(small.data_ptr()), C.c_void_p(small.data_ptr()), 8, C.byref(h))
    assert rc == -1 and not h


def test_gzip_input_goes_through_the_device_ingest(tmp_path, knobs, monkeypatch):
    """feature.py:76-91 reads *.gz through one `pigz -dc` stream; here the text is inflated into an in-memory file
    (pg_inflate_to_memfd) and then takes the same pieces, copies and placement kernel as a plain file -- the golden gzip input,
    a multi-member gzip file and a larger synthetic one, against the host ingest; a text that may not be parked in memory
    (PG_INFLATE_MAX_BYTES) falls back to the host ingest, with the same stream"""
    L = knobs
    calls = []
    real = ReadStream._ingest_plain_to_device.__func__
    monkeypatch.setattr(ReadStream, "_ingest_plain_to_device", classmethod(lambda cls, L_, reads, *a: (calls.append(reads), real(cls, L_, reads, *a))[1]))
    golden = os.path.join(GOLDEN, "polya.fq.gz")
    big = str(tmp_path / "big.fq.gz")
    cfg = synth.SynthConfig(n_pairs=3000, n_barcodes=40, n_genomes=2, genome_len=20_000, fragment=2_000, n_rate=0.1, seed=5)
    plain = str(tmp_path / "big.fq")
    synth.write_fastq(synth.generate(cfg), cfg, plain)
    text = open(plain, "rb").read()
    with gzip.open(big, "wb") as f:
        f.write(text)
    members = str(tmp_path / "members.fq.gz")              # two gzip members back to back (what `cat a.gz b.gz` makes)
    cut = text.index(b"\n@", len(text) // 2) + 1
    open(members, "wb").write(gzip.compress(text[:cut]) + gzip.compress(text[cut:]))
    for path in (golden, big, members):
        for threads, piece in ((1, None), (4, "4096")):
            L.pg_set_ingest_threads(threads)
            if piece:
                os.environ["PG_INGEST_PIECE"] = piece
            else:
                os.environ.pop("PG_INGEST_PIECE", None)
            n = len(calls)
            got = ReadStream.from_fastq(path, device=DEV)
            assert len(calls) == n + 1 and calls[-1].startswith("/proc/self/fd/")
            _same(got, ReadStream.from_fastq(path))
    _same(ReadStream.from_fastq(members, device=DEV), ReadStream.from_fastq(plain))
    os.environ.pop("PG_INGEST_PIECE", None)
    assert not [f for f in os.listdir("/proc/self/fd") if os.path.realpath(f"/proc/self/fd/{f}").startswith("/memfd:pg_inflate")]     # descriptors closed
    monkeypatch.setenv("PG_INFLATE_MAX_BYTES", "1000")
    n = len(calls)
    _same(ReadStream.from_fastq(big, device=DEV), ReadStream.from_fastq(plain))
    assert len(calls) == n


def _split_pairs(interleaved: str, r1: str, r2: str, mutate=None):
    """an interleaved FASTQ file as -1 / -2 files (records alternate); ``mutate(i, rec1, rec2)`` may change pair i"""
    lines = open(interleaved).read().splitlines(keepends=True)
    with open(r1, "w") as a, open(r2, "w") as b:
        for i in range(0, len(lines) - 7, 8):
            x, y = lines[i:i + 4], lines[i + 4:i + 8]
            if mutate:
                x, y = mutate(i // 8, x, y)
            a.writelines(x); b.writelines(y)


def test_paired_files_go_through_the_device_ingest(tmp_path, knobs, monkeypatch):
    """-1 / -2 input (count_tnf.cpp:174-231 + the low-quality plane of feature.py:76-83) with the copy to the GPU inside
    (pg_ingest_fastq_pair_device + pg_ingest_place_pair): same stream, runs, counters and planes as the host ingest, for pieces of
    a few hundred bytes up to one piece, one thread and several; the golden pairs; pairs that do not match (skipped, their reads
    behind the last run), low qualities, soft-masked bases, an R2 that is shorter; gzip pairs by way of the in-memory files"""
    L = knobs
    took = []                                                             # did the device form take the input (or hand it back)?
    real = ReadStream._ingest_pair_to_device.__func__
    monkeypatch.setattr(ReadStream, "_ingest_pair_to_device", classmethod(lambda cls, *a: (lambda r: (took.append(r is not None), r)[1])(real(cls, *a))))
    for stem in ("pair", "pairq"):
        r1, r2 = os.path.join(GOLDEN, f"{stem}_R1.fq"), os.path.join(GOLDEN, f"{stem}_R2.fq")
        L.pg_set_ingest_threads(1)
        want = ReadStream.from_fastq(r1, r2)
        for threads, piece in ((1, None), (3, "200"), (4, "4096")):
            L.pg_set_ingest_threads(threads)
            if piece:
                os.environ["PG_INGEST_PIECE"] = piece
            else:
                os.environ.pop("PG_INGEST_PIECE", None)
            _same(ReadStream.from_fastq(r1, r2, device=DEV), want)
    assert took == [True] * 6
    # a larger pair with everything in it
    cfg = synth.SynthConfig(n_pairs=20_000, n_barcodes=97, n_genomes=2, genome_len=30_000, fragment=3_000, n_rate=0.1, unbarcoded=0.05, seed=11)
    fq = str(tmp_path / "i.fq")
    synth.write_fastq(synth.generate(cfg), cfg, fq)
    rng = np.random.RandomState(3)

    def mutate(i, x, y):
        u = rng.rand()
        if u < 0.02:                                                      # names differ: the pair is skipped
            y = [y[0].replace("@", "@x", 1)] + y[1:]
        elif u < 0.3:                                                     # low qualities in one read
            q = list(x[3].rstrip("\n"))
            for j in rng.randint(0, len(q), size=5):
                q[j] = "#"
            x = x[:3] + ["".join(q) + "\n"]
        elif u < 0.35:                                                    # a soft-masked stretch
            sq = y[1].rstrip("\n")
            y = [y[0], sq[:10] + sq[10:40].lower() + sq[40:] + "\n"] + y[2:]
        return x, y

    r1, r2 = str(tmp_path / "a_R1.fq"), str(tmp_path / "a_R2.fq")
    _split_pairs(fq, r1, r2, mutate)
    L.pg_set_ingest_threads(1)
    os.environ.pop("PG_INGEST_PIECE", None)
    os.environ["PANGAEA_INGEST_ON_HOST"] = "1"
    want = ReadStream.from_fastq(r1, r2, device=DEV)                      # (host ingest + one copy)
    os.environ.pop("PANGAEA_INGEST_ON_HOST")
    assert want.valid_lowq is not None and want.valid_lower is not None and want.n_unpaired > 0
    want_host = ReadStream.from_fastq(r1, r2)
    for threads, piece in ((1, None), (4, "30000"), (8, "1000")):
        L.pg_set_ingest_threads(threads)
        if piece:
            os.environ["PG_INGEST_PIECE"] = piece
        else:
            os.environ.pop("PG_INGEST_PIECE", None)
        _same(ReadStream.from_fastq(r1, r2, device=DEV), want_host)
    assert took[-3:] == [True, True, True]
    # an R2 that ends early: what the threads cannot pair is the serial tail's
    short2 = str(tmp_path / "short_R2.fq")
    lines = open(r2).read().splitlines(keepends=True)
    open(short2, "w").writelines(lines[:len(lines) // 2 // 4 * 4 - 2])
    L.pg_set_ingest_threads(4)
    os.environ["PG_INGEST_PIECE"] = "50000"
    _same(ReadStream.from_fastq(r1, short2, device=DEV), ReadStream.from_fastq(r1, short2))
    # gzip pairs: inflated side by side into in-memory files, then the same ingest
    import gzip as gz
    g1, g2 = str(tmp_path / "a_R1.fq.gz"), str(tmp_path / "a_R2.fq.gz")
    for src, dst in ((r1, g1), (r2, g2)):
        with gz.open(dst, "wb") as f:
            f.write(open(src, "rb").read())
    _same(ReadStream.from_fastq(g1, g2, device=DEV), want_host)
    _same(ReadStream.from_fastq(g1, r2, device=DEV), want_host)
    assert took[-3:] == [True, True, True]
    os.environ.pop("PG_INGEST_PIECE", None)
    assert not [f for f in os.listdir("/proc/self/fd") if os.path.realpath(f"/proc/self/fd/{f}").startswith("/memfd:pg_inflate")]
    # staging arrays that are too small are refused
    n1, n2 = os.path.getsize(r1), os.path.getsize(r2)
    cap = int(L.pg_ingest_pair_staging_words(n1, n2))
    sc = torch.empty(cap, dtype=torch.int64, device=DEV); sv = torch.empty(cap, dtype=torch.int32, device=DEV); sq = torch.empty(cap, dtype=torch.int32, device=DEV)
    h = C.c_void_p()
    rc = L.pg_ingest_fastq_pair_device(r1.encode(), r2.encode(), n1, n2, C.c_void_p(sc.data_ptr()), C.c_void_p(sv.data_ptr()), C.c_void_p(sq.data_ptr()), cap - 1, C.byref(h))
    assert rc != 0 and not h
