"""Bin writer (pg_extract_reads) against the files the reference's extract_reads binary wrote (tests/golden/bins_*)
and, where oracle/_ref exists, against that binary live on randomised cluster assignments."""
import os
import random
import subprocess

import pytest

from oracle import oracle
from pangaea_amd import _lib
from pangaea_amd.binwriter import extract_reads
from pangaea_amd.clustering import write_clusters_tsv

from .conftest import GOLDEN


def _inputs(spec):
    if "i" in spec:
        return os.path.join(GOLDEN, spec["i"]), None
    return os.path.join(GOLDEN, spec["1"]), os.path.join(GOLDEN, spec["2"])


def test_golden_bins(manifest, tmp_path):
    for case in manifest["bin_writer"]:
        src = os.path.join(GOLDEN, case["dir"])
        out = tmp_path / case["dir"]
        out.mkdir()
        r1, r2 = _inputs(case["input"])
        n = extract_reads(r1, r2, os.path.join(src, "clusters.tsv"), str(out / "cluster"))
        assert sorted(os.listdir(out)) == case["files"]
        total = 0
        for fn in case["files"]:
            with open(os.path.join(src, fn), "rb") as f, open(out / fn, "rb") as g:
                want = f.read()
                assert g.read() == want, fn
            if fn.endswith(".barcode"):
                total += want.count(b"\n")
        assert n == total


def test_errors_are_reported(tmp_path):
    with pytest.raises(_lib.PangaeaError):
        extract_reads(os.path.join(GOLDEN, "stlfr.fq"), None, str(tmp_path / "missing.tsv"), str(tmp_path / "c"))
    tsv = tmp_path / "c.tsv"
    tsv.write_text("0\tAAAA\n")
    with pytest.raises(_lib.PangaeaError):
        extract_reads("/nonexistent.fq", None, str(tsv), str(tmp_path / "c"))
    with pytest.raises(_lib.PangaeaError):
        extract_reads(os.path.join(GOLDEN, "stlfr.fq"), None, str(tsv), str(tmp_path / "no_such_dir" / "c"))


@pytest.mark.skipif(oracle.ref_tool("extract_reads") is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name", ["tenx_mixed.fq", "tenx_clean.fq.gz", "stlfr.fq", "tenx_crlf.fq", "pair"])
def test_live_against_reference_binary(name, tmp_path):
    rng = random.Random(len(name))
    spec = {"1": "pair_R1.fq", "2": "pair_R2.fq"} if name == "pair" else {"i": name}
    r1, r2 = _inputs(spec)
    barcodes = sorted({n for n in oracle.Reads(r1, r2).names if n})
    labels = [rng.choice([-1, 0, 1, 2, 5]) for _ in barcodes]
    tsv = str(tmp_path / "clusters.tsv")
    write_clusters_tsv(tsv, labels, barcodes)
    a, b = tmp_path / "ref", tmp_path / "mine"
    a.mkdir(); b.mkdir()
    flags = ["-i", r1] if r2 is None else ["-1", r1, "-2", r2]
    subprocess.run([oracle.ref_tool("extract_reads")] + flags + ["-c", tsv, "-o", str(a / "cluster")], check=True, stdout=subprocess.DEVNULL)
    extract_reads(r1, r2, tsv, str(b / "cluster"))
    assert sorted(os.listdir(a)) == sorted(os.listdir(b))
    for fn in os.listdir(a):
        assert (a / fn).read_bytes() == (b / fn).read_bytes(), fn


@pytest.mark.parametrize("style", ["10x", "stlfr"])
def test_threaded_bin_writer_writes_the_same_bytes(style, tmp_path):
    """a file big enough for the threaded form (ranges streamed twice: sizes, then pwrite at exact offsets) gives, for any
    thread count and reader block size, exactly the files of the sequential loop -- and of the reference's extract_reads"""
    import numpy as np
    from pangaea_amd import synth
    cfg = synth.SynthConfig(n_pairs=7000, n_barcodes=60, n_genomes=2, genome_len=30_000, fragment=5_000, unbarcoded=0.1, seed=3)
    stream = synth.generate(cfg)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(stream, cfg, fq, style=style)
    lines = open(fq).read().splitlines(keepends=True)
    open(fq, "w").write("".join(lines[:-3]))                       # the last record is cut short: never written
    barcodes = sorted({n for n in oracle.Reads(fq).names if n})
    rng = random.Random(7)
    tsv = str(tmp_path / "clusters.tsv")
    write_clusters_tsv(tsv, [rng.choice([-1, 0, 1, 2, 3, 9]) for _ in barcodes], barcodes)
    L = _lib.load()
    outs = {}
    try:
        for tag, threads, block in (("serial", 1, None), ("t4", 4, None), ("t7_small_blocks", 7, "700")):
            L.pg_set_ingest_threads(threads)
            if block:
                os.environ["PG_INGEST_BLOCK"] = block
            d = tmp_path / tag
            d.mkdir()
            n = extract_reads(fq, None, tsv, str(d / "cluster"))
            outs[tag] = (n, {fn: (d / fn).read_bytes() for fn in sorted(os.listdir(d))})
    finally:
        L.pg_set_ingest_threads(0)
        os.environ.pop("PG_INGEST_BLOCK", None)
    assert outs["serial"][0] > 0 and len(outs["serial"][1]) == 10
    assert outs["t4"] == outs["serial"] and outs["t7_small_blocks"] == outs["serial"]
    if oracle.ref_tool("extract_reads") is not None:
        d = tmp_path / "ref"
        d.mkdir()
        subprocess.run([oracle.ref_tool("extract_reads"), "-i", fq, "-c", tsv, "-o", str(d / "cluster")], check=True, stdout=subprocess.DEVNULL)
        assert {fn: (d / fn).read_bytes() for fn in sorted(os.listdir(d))} == outs["serial"][1]


def test_threaded_paired_bin_writer_writes_the_same_bytes(tmp_path):
    """-1 / -2 files big enough for the threaded form: the same cluster files as the sequential loop for any thread count and
    block size -- pairs whose names or barcodes differ are left out, the last R1 record is cut short, R2 is longer -- and as
    the reference's extract_reads on the same two files"""
    from pangaea_amd import synth
    cfg = synth.SynthConfig(n_pairs=7000, n_barcodes=60, n_genomes=2, genome_len=30_000, fragment=5_000, unbarcoded=0.1, seed=5)
    fq = str(tmp_path / "i.fq")
    synth.write_fastq(synth.generate(cfg), cfg, fq)
    lines = open(fq).read().splitlines()
    recs = [lines[i:i + 4] for i in range(0, len(lines), 4)]
    p1, p2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    with open(p1, "w") as o1, open(p2, "w") as o2:
        for i, (a, b) in enumerate(zip(recs[0::2], recs[1::2])):
            if i % 19 == 0:
                b[0] = b[0].replace("@", "@y", 1)
            if i % 23 == 0 and "BX:Z:" in b[0]:
                b[0] = b[0].replace("BX:Z:A", "BX:Z:C").replace("BX:Z:G", "BX:Z:T")
            o2.write("\n".join(b) + "\n")
            if i < len(recs) // 2 - 9:
                o1.write("\n".join(a) + "\n")
            elif i == len(recs) // 2 - 9:
                o1.write("\n".join(a[:3]) + "\n")                   # cut short: never written
    barcodes = sorted({n for n in oracle.Reads(fq).names if n})
    rng = random.Random(8)
    tsv = str(tmp_path / "clusters.tsv")
    write_clusters_tsv(tsv, [rng.choice([-1, 0, 1, 2, 5]) for _ in barcodes], barcodes)
    L = _lib.load()
    outs = {}
    try:
        for tag, threads, block in (("serial", 1, None), ("t3", 3, None), ("t6_small_blocks", 6, "500")):
            L.pg_set_ingest_threads(threads)
            if block:
                os.environ["PG_INGEST_BLOCK"] = block
            d = tmp_path / tag
            d.mkdir()
            n = extract_reads(p1, p2, tsv, str(d / "cluster"))
            outs[tag] = (n, {fn: (d / fn).read_bytes() for fn in sorted(os.listdir(d))})
    finally:
        L.pg_set_ingest_threads(0)
        os.environ.pop("PG_INGEST_BLOCK", None)
    assert outs["serial"][0] > 1000 and len(outs["serial"][1]) == 8
    assert outs["t3"] == outs["serial"] and outs["t6_small_blocks"] == outs["serial"]
    if oracle.ref_tool("extract_reads") is not None:
        d = tmp_path / "ref"
        d.mkdir()
        subprocess.run([oracle.ref_tool("extract_reads"), "-1", p1, "-2", p2, "-c", tsv, "-o", str(d / "cluster")], check=True, stdout=subprocess.DEVNULL)
        assert {fn: (d / fn).read_bytes() for fn in sorted(os.listdir(d))} == outs["serial"][1]
