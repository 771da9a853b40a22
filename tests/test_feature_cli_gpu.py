"""The reference-facing interfaces on the GPU: ``Feature`` (src/feature.py) and the two CLI tools."""
import argparse
import gzip
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from .conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _args(tmp_path, **kw):
    d = dict(reads1="", reads2="", interleaved_reads="", output=str(tmp_path / "out"), min_length=2000, kmer=15,
             tnf_kmer=4, window_size=10, vector_size=400, threads=4)
    d.update(kw)
    os.makedirs(d["output"], exist_ok=True)
    return argparse.Namespace(**d)


def test_feature_matches_reference_flow(tmp_path):
    """names / abundance / tnf equal what the reference's Feature returns: pandas' reading of the CSVs written by its
    binaries (goldens), and the cache files are in place for -st 2,3,4 resumes"""
    from pangaea_amd.feature import Feature
    args = _args(tmp_path, interleaved_reads=os.path.join(GOLDEN, "tenx_clean.fq.gz"))
    feat = Feature(args, ROOT)
    names, abd, tnf = feat.extract_features()
    ref_t = pd.read_csv(os.path.join(GOLDEN, "tenx_clean.tnf.k4.l2000.csv"), header=None)
    ref_a = pd.read_csv(os.path.join(GOLDEN, "tenx_clean.abd.k15.w10.v400.l2000.csv"), header=None)
    assert (names == ref_t[0].to_numpy()).all() and names.dtype == ref_t[0].to_numpy().dtype
    assert np.array_equal(tnf, ref_t.drop(columns=0).to_numpy()) and tnf.dtype == np.int64
    assert np.array_equal(abd, ref_a.drop(columns=0).to_numpy()) and abd.dtype == np.int64
    fdir = os.path.join(args.output, "1.features")
    for fn in ("tnf.m2000.gz", "tnf.m2000.pkl", "abundance.k15.v400.w10.m2000.gz", "abundance.k15.v400.w10.m2000.pkl", "feature_finished"):
        assert os.path.isfile(os.path.join(fdir, fn)), fn
    with gzip.open(os.path.join(fdir, "tnf.m2000.gz"), "rb") as f, open(os.path.join(GOLDEN, "tenx_clean.tnf.k4.l2000.csv"), "rb") as g:
        assert f.read() == g.read()
    # resume: a new object loads the caches without touching the reads
    args2 = _args(tmp_path, interleaved_reads="/nonexistent.fq")
    n2, a2, t2 = Feature(args2, ROOT).load_features()
    assert (n2 == names).all() and np.array_equal(a2, abd) and np.array_equal(t2, tnf)
    n3, a3, t3 = Feature(args2, ROOT).extract_features()
    assert (n3 == names).all() and np.array_equal(a3, abd) and np.array_equal(t3, tnf)


def test_feature_paired_inputs_and_exponent_counts(tmp_path):
    from pangaea_amd.feature import Feature
    args = _args(tmp_path, reads1=os.path.join(GOLDEN, "pair_R1.fq"), reads2=os.path.join(GOLDEN, "pair_R2.fq"),
                 min_length=100, kmer=15, window_size=1, vector_size=6)
    names, abd, tnf = Feature(args, ROOT).extract_features()
    ref_t = pd.read_csv(os.path.join(GOLDEN, "pair.tnf.k4.l100.csv"), header=None)
    ref_a = pd.read_csv(os.path.join(GOLDEN, "pair.abd.k15.w1.v6.l100.csv"), header=None)
    assert (names == ref_t[0].to_numpy()).all()
    assert np.array_equal(tnf, ref_t.drop(columns=0).to_numpy()) and np.array_equal(abd, ref_a.drop(columns=0).to_numpy())
    # counts >= 1e6 come back the way pandas reads "%g" text
    args = _args(tmp_path / "p", interleaved_reads=os.path.join(GOLDEN, "polya.fq.gz"), min_length=1000)
    os.makedirs(args.output, exist_ok=True)
    names, abd, tnf = Feature(args, ROOT).extract_features()
    ref_t = pd.read_csv(os.path.join(GOLDEN, "polya.tnf.k4.l1000.csv"), header=None)
    assert tnf.dtype == np.float64 and np.array_equal(tnf, ref_t.drop(columns=0).to_numpy())


def test_feature_follows_jellyfish_rules_for_the_table(tmp_path, monkeypatch):
    """-1/-2 input: the multiplicity table leaves out bases below '?' (jellyfish --min-qual-char=?, feature.py:76-83) while the
    rows do not look at qualities; soft-masked reads: lower-case bases count for the table, reset the rows' counters.
    Goldens = the reference's count_kmer / count_tnf on the same files, given the dump those rules produce."""
    from oracle import oracle
    from pangaea_amd import kmer
    from pangaea_amd.feature import Feature
    from pangaea_amd.reads import ReadStream
    r1, r2 = os.path.join(GOLDEN, "pairq_R1.fq"), os.path.join(GOLDEN, "pairq_R2.fq")
    for k, w, v in ((15, 1, 6), (21, 1, 6), (9, 2, 50)):
        args = _args(tmp_path / f"q{k}", reads1=r1, reads2=r2, min_length=100, kmer=k, window_size=w, vector_size=v)
        names, abd, tnf = Feature(args, ROOT).extract_features()
        ref_a = pd.read_csv(os.path.join(GOLDEN, f"pairq.abd.k{k}.w{w}.v{v}.l100.csv"), header=None)
        ref_t = pd.read_csv(os.path.join(GOLDEN, "pairq.tnf.k4.l100.csv"), header=None)
        assert (names == ref_a[0].to_numpy()).all()
        assert np.array_equal(abd, ref_a.drop(columns=0).to_numpy()) and np.array_equal(tnf, ref_t.drop(columns=0).to_numpy())
    # the table itself, in every counting form, is the dump (rows and fused lookups are dropped for such streams)
    s = ReadStream.from_fastq(r1, r2, device="cuda:0")
    assert s.valid_lowq is not None and not s.rows_inside_table
    want = oracle.Table.from_dump(os.path.join(GOLDEN, "pairq.k21.dump"), 21).items()
    plan = kmer.Plan(s.rows(100), "cuda:0")
    for t in (kmer.KmerTable.with_slots(21, "cuda:0", 16, 0).count(s),                                   # direct
              kmer.KmerTable.with_slots(21, "cuda:0", 18, 6).count(s, rows=plan, emit=(1, 6)),          # bucketed (fusion dropped)
              kmer.KmerTable.mini_with_slots(21, "cuda:0", 18, 10).count(s, rows=plan, emit=(1, 6)),     # super-k-mers
              kmer.count_kmers(s, 21, rows=plan, emit=(1, 6))):
        assert t._emitted is None and t._records is None
        assert all(np.array_equal(x, y) for x, y in zip(t.items(), want))
    # the unmasked table would be another one
    monkeypatch.setattr(ReadStream, "table_valid", lambda self, lowercase_is_base=True: self.valid)
    assert not all(np.array_equal(x, y) for x, y in zip(kmer.KmerTable.with_slots(21, "cuda:0", 16, 0).count(s).items(), want))
    monkeypatch.undo()
    # soft-masked reads, interleaved
    for k in (15, 21):
        args = _args(tmp_path / f"s{k}", interleaved_reads=os.path.join(GOLDEN, "soft.fq"), min_length=100, kmer=k, window_size=1, vector_size=6)
        names, abd, tnf = Feature(args, ROOT).extract_features()
        ref_a = pd.read_csv(os.path.join(GOLDEN, f"soft.abd.k{k}.w1.v6.l100.csv"), header=None)
        ref_t = pd.read_csv(os.path.join(GOLDEN, "soft.tnf.k4.l100.csv"), header=None)
        assert np.array_equal(abd, ref_a.drop(columns=0).to_numpy()) and np.array_equal(tnf, ref_t.drop(columns=0).to_numpy())
    monkeypatch.setenv("PANGAEA_LOWERCASE_IS_BASE", "0")             # the strict table gives other rows on this input
    args = _args(tmp_path / "strict", interleaved_reads=os.path.join(GOLDEN, "soft.fq"), min_length=100, kmer=15, window_size=1, vector_size=6)
    _, abd_strict, _ = Feature(args, ROOT).extract_features()
    assert not np.array_equal(abd_strict, pd.read_csv(os.path.join(GOLDEN, "soft.abd.k15.w1.v6.l100.csv"), header=None).drop(columns=0).to_numpy())


def _run(tool, *argv):
    return subprocess.run([sys.executable, os.path.join(ROOT, "pangaea_amd", "bin", tool), *argv], capture_output=True, text=True)


def test_cli_tools_write_the_reference_files(tmp_path):
    out = str(tmp_path / "t.gz")
    r = _run("count_tnf", "-i", os.path.join(GOLDEN, "stlfr.fq"), "-k", "4", "-l", "100", "-t", "3", "-o", out)
    assert r.returncode == 0, r.stderr
    with gzip.open(out, "rb") as f, open(os.path.join(GOLDEN, "stlfr.tnf.k4.l100.csv"), "rb") as g:
        assert f.read() == g.read()
    # count_kmer with an existing (holed) dump: reference loader semantics
    out = str(tmp_path / "a.gz")
    r = _run("count_kmer", "-i", os.path.join(GOLDEN, "tenx_mixed.fq"), "-g", os.path.join(GOLDEN, "tenx_mixed.k21.holes.dump"),
             "-k", "21", "-w", "10", "-v", "400", "-l", "0", "-o", out)
    assert r.returncode == 0, r.stderr
    with gzip.open(out, "rb") as f, open(os.path.join(GOLDEN, "tenx_mixed.abd.k21.w10.v400.l0.holes.csv"), "rb") as g:
        assert f.read() == g.read()
    # without a dump file the multiplicities are counted on the GPU (what jellyfish would have dumped)
    out = str(tmp_path / "b.gz")
    r = _run("count_kmer", "-1", os.path.join(GOLDEN, "pair_R1.fq"), "-2", os.path.join(GOLDEN, "pair_R2.fq"),
             "-g", str(tmp_path / "absent.dump"), "-k", "15", "-w", "1", "-v", "6", "-l", "100", "-o", out)
    assert r.returncode == 0, r.stderr
    with gzip.open(out, "rb") as f, open(os.path.join(GOLDEN, "pair.abd.k15.w1.v6.l100.csv"), "rb") as g:
        assert f.read() == g.read()
    # bad usage exits 1 like cmdline.h
    assert _run("count_tnf", "-i", "x.fq").returncode == 1
    assert _run("count_kmer", "-i", "x.fq", "-o", out).returncode == 1
    assert _run("count_tnf", "-1", "only_one.fq", "-o", out).returncode == 1


@pytest.mark.parametrize("style,k,clusters", [("stlfr", 21, 30), ("hybrid", 15, 40)])
def test_stlfr_and_hybrid_style_inputs(style, k, clusters, tmp_path):
    """BASELINE configs 4 and 5 in miniature: raw stLFR headers (barcode in the read name, 0_0_0 = none) and hybrid mode
    (long-read names as barcodes) through Feature on the GPU, against the oracle's reading of the same file"""
    from oracle import oracle
    from pangaea_amd import synth
    from pangaea_amd.feature import Feature
    cfg = synth.SynthConfig(n_pairs=5000, n_barcodes=60, n_genomes=3, genome_len=60_000, fragment=15_000, seed=11)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(synth.generate(cfg), cfg, fq, style=style)
    args = _args(tmp_path, interleaved_reads=fq, kmer=k, min_length=2000)
    names, abd, tnf = Feature(args, ROOT).extract_features()
    rd = oracle.Reads(fq)
    assert rd.mode == ("stLFR" if style == "stlfr" else "10x")
    table = oracle.Table(k, threads=4).count(rd.all_seq())
    onames, otnf, oabd = rd.features(2000, k_tnf=4, k_abd=k, table=table, window=10, vsize=400, threads=4)
    assert list(names) == onames and len(onames) == 60
    assert np.array_equal(tnf, otnf) and np.array_equal(abd, oabd)
    if style == "stlfr":
        assert names[0] == "1_1_1"
    else:
        assert names[0] == "lr_0000000"


def test_packed_stream_cache_feeds_a_second_pass(tmp_path, monkeypatch):
    """PANGAEA_STREAM_CACHE=1: the first pass leaves the packed read stream next to the feature caches; a second pass with
    other parameters (here another k) reads it instead of parsing the FASTQ again, and gives the reference's rows"""
    from pangaea_amd.feature import Feature
    from pangaea_amd.reads import ReadStream
    monkeypatch.setenv("PANGAEA_STREAM_CACHE", "1")
    args = _args(tmp_path, interleaved_reads=os.path.join(GOLDEN, "tenx_clean.fq.gz"))
    names, abd, tnf = Feature(args, ROOT).extract_features()
    cache = os.path.join(args.output, "1.features", "reads.r0of1.pgstream")
    assert os.path.isfile(cache)
    calls = []
    real = ReadStream.from_fastq.__func__
    monkeypatch.setattr(ReadStream, "from_fastq", classmethod(lambda cls, *a, **k: calls.append(a) or real(cls, *a, **k)))
    args21 = _args(tmp_path, interleaved_reads=os.path.join(GOLDEN, "tenx_clean.fq.gz"), kmer=21, window_size=1, vector_size=6,
                   min_length=1000, tnf_kmer=5)
    n2, a2, t2 = Feature(args21, ROOT).extract_features()
    assert calls == []                                               # no FASTQ parse
    ref_a = pd.read_csv(os.path.join(GOLDEN, "tenx_clean.abd.k21.w1.v6.l1000.csv"), header=None)
    ref_t = pd.read_csv(os.path.join(GOLDEN, "tenx_clean.tnf.k5.l1000.csv"), header=None)
    assert (n2 == ref_a[0].to_numpy()).all()
    assert np.array_equal(t2, ref_t.drop(columns=0).to_numpy()) and np.array_equal(a2, ref_a.drop(columns=0).to_numpy())


def test_feature_on_paired_files_through_the_threaded_reader(tmp_path, monkeypatch):
    """-1 / -2 files large enough for the threaded paired reader (records cut by byte ranges of R1, located in R2): names,
    abundance and TNF through ``Feature`` equal the oracle's reading of the same two files under jellyfish's rules (bases of
    quality below '?' leave the table, skipped pairs still feed it), serial and threaded ingest give the same matrices"""
    from oracle import oracle
    from pangaea_amd import _lib, synth
    from pangaea_amd.feature import Feature
    rs = np.random.RandomState(3)
    cfg = synth.SynthConfig(n_pairs=5000, n_barcodes=23, n_genomes=2, genome_len=30_000, fragment=6_000, n_rate=0.2, unbarcoded=0.03)
    fq = str(tmp_path / "i.fq")
    synth.write_fastq(synth.generate(cfg), cfg, fq)
    lines = open(fq).read().splitlines()
    recs = [lines[i:i + 4] for i in range(0, len(lines), 4)]
    p1, p2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    with open(p1, "w") as o1, open(p2, "w") as o2:
        for i, (a, b) in enumerate(zip(recs[0::2], recs[1::2])):
            a[3] = "".join(rs.choice(list("#5?FI"), size=len(a[1]), p=[0.03, 0.07, 0.1, 0.3, 0.5]))
            b[3] = "".join(rs.choice(list("#5?FI"), size=len(b[1]), p=[0.03, 0.07, 0.1, 0.3, 0.5]))
            if i % 29 == 0:
                b[0] = b[0].replace("@", "@x", 1)
            o1.write("\n".join(a) + "\n")
            o2.write("\n".join(b) + "\n")
    assert os.path.getsize(p1) > (1 << 16) * 4
    rd = oracle.Reads(p1, p2)
    table = oracle.Table(15, threads=4).count(np.frombuffer(rd.all_seq(), dtype=np.uint8))
    names_o, tnf_o, abd_o = rd.features(1000, k_tnf=4, k_abd=15, table=table, window=2, vsize=100, threads=4)
    out = {}
    try:
        for T in (4, 1):
            _lib.load().pg_set_ingest_threads(T)
            args = _args(tmp_path / f"t{T}", reads1=p1, reads2=p2, min_length=1000, kmer=15, window_size=2, vector_size=100)
            out[T] = Feature(args, ROOT).extract_features()
    finally:
        _lib.load().pg_set_ingest_threads(0)
    names, abd, tnf = out[4]
    assert list(names) == list(names_o) and len(names) > 10
    assert np.array_equal(abd, abd_o) and np.array_equal(tnf, tnf_o)
    assert all(np.array_equal(x, y) for x, y in zip(out[4], out[1]))
