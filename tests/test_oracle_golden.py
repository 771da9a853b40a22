"""The oracle (oracle/pangaea_oracle.c) against the reference's own outputs.

tests/golden/*.csv were written by the reference's count_tnf / count_kmer binaries
(tests/golden/make_goldens.py); the oracle must reproduce every byte, including the %g number
formatting, the row order and the set of surviving rows.
"""
import gzip
import hashlib
import json
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import oracle

from .conftest import GOLDEN

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _CASES = json.load(_f)["cases"]


def _reads(spec):
    r1 = os.path.join(GOLDEN, spec.get("i") or spec["1"])
    r2 = os.path.join(GOLDEN, spec["2"]) if "2" in spec else None
    return oracle.Reads(r1, r2)


def _csv_bytes(tmp_path, names, mat):
    p = str(tmp_path / "o.csv.gz")
    oracle.write_csv_gz(p, names, mat)
    with gzip.open(p, "rb") as f:
        return f.read()


@pytest.mark.parametrize("case", [c for c in _CASES if c["tool"] == "count_tnf"], ids=lambda c: c["expect"])
def test_tnf_matches_reference_binary(case, tmp_path):
    rd = _reads(case["input"])
    names, tnf, _ = rd.features(case["min_len"], k_tnf=case["k"])
    with open(os.path.join(GOLDEN, case["expect"]), "rb") as f:
        assert _csv_bytes(tmp_path, names, tnf) == f.read()


@pytest.mark.parametrize("case", [c for c in _CASES if c["tool"] == "count_kmer"], ids=lambda c: c["expect"])
def test_abundance_matches_reference_binary(case, tmp_path):
    rd = _reads(case["input"])
    table = oracle.Table.from_dump(os.path.join(GOLDEN, case["dump"]), case["k"])
    names, _, abd = rd.features(case["min_len"], k_tnf=None, k_abd=case["k"], table=table,
                                window=case["window"], vsize=case["vsize"])
    with open(os.path.join(GOLDEN, case["expect"]), "rb") as f:
        assert _csv_bytes(tmp_path, names, abd) == f.read()


def test_exact_counter_equals_dump_when_complete():
    """the oracle's own counter (jellyfish stand-in) and its dump -> reference-style loader round trip"""
    for case in _CASES:
        if case["tool"] != "count_kmer" or case["holes"]:
            continue
        rd = _reads(case["input"])
        direct = oracle.Table(case["k"], threads=3).count(rd.all_seq(), lowercase_is_base=case.get("jellyfish_rules", False))
        loaded = oracle.Table.from_dump(os.path.join(GOLDEN, case["dump"]), case["k"])
        k1, v1 = direct.items()
        k2, v2 = loaded.items()
        assert np.array_equal(k1, k2) and np.array_equal(v1, v2), case["expect"]


def test_jellyfish_rules_of_the_paired_branch_and_of_soft_masked_reads():
    """what jellyfish sees differs from what count_tnf / count_kmer see in two ways (feature.py:76-94): --min-qual-char=? on
    paired files turns bases below '?' into N (all_seq carries that), and lower-case bases count"""
    rd = _reads({"1": "pairq_R1.fq", "2": "pairq_R2.fq"})
    seen = rd.all_seq()
    raw = b"".join(ln.rstrip(b"\n") + b"N" for fn in ("pairq_R1.fq", "pairq_R2.fq")
                   for i, ln in enumerate(open(os.path.join(GOLDEN, fn), "rb")) if i % 4 == 1)
    qual = b"".join(ln.rstrip(b"\n") + b"I" for fn in ("pairq_R1.fq", "pairq_R2.fq")
                    for i, ln in enumerate(open(os.path.join(GOLDEN, fn), "rb")) if i % 4 == 3)
    assert len(seen) == len(raw) == len(qual)
    want = bytes(ord("N") if q < ord("?") else c for c, q in zip(raw, qual))
    assert seen == want and seen != raw and b"N" * 45 in seen          # (one read has no trusted base at all)
    masked = oracle.Table(15).count(seen)
    unmasked = oracle.Table(15).count(raw)
    assert 0 < len(masked) < len(unmasked)
    sm = _reads({"i": "soft.fq"})
    strict, lenient = oracle.Table(15).count(sm.all_seq()), oracle.Table(15).count(sm.all_seq(), lowercase_is_base=True)
    assert int(strict.items()[1].sum()) < int(lenient.items()[1].sum())
    assert all(np.array_equal(a, b) for a, b in zip(lenient.items(), oracle.Table(15).count(sm.all_seq().upper()).items()))


def test_tnf_column_order_anchor(manifest):
    for k, ref in manifest["tnf_column_anchors"].items():
        cols = oracle.tnf_columns(int(k))
        assert len(cols) == ref["ncols"]
        hdr = ",".join(oracle.code_to_kmer(int(c), int(k)) for c in cols)
        assert hashlib.sha256(hdr.encode()).hexdigest()[:16] == ref["sha256_16"]
    assert [oracle.code_to_kmer(int(c), 2) for c in oracle.tnf_columns(2)] == \
        ["AA", "AC", "AT", "AG", "CA", "CC", "CG", "TA", "TC", "GC"]
    assert oracle.tnf_ncols(4) == 136


def test_run_quirks_are_reproduced():
    """append-then-compare grouping: first pair of the file is lost, every run ends with the first pair
    of the next one, and the unbarcoded runs are dropped (count_tnf.cpp:245-274)."""
    rd = oracle.Reads(os.path.join(GOLDEN, "tenx_len.fq"))
    assert rd.names == ["", "AAAAAAAA", "CCCCCCCC", "GGGGGGGG", "TTTTTTTT"]
    # 3,4,5,6 pairs of 2x(20+1) chars -> runs of 1, 3, 4, 5 and the trailing 5 pairs
    assert list(np.diff(rd.seq_off)) == [42, 3 * 42, 4 * 42, 5 * 42, 5 * 42]
    assert rd.surviving(168) == [3, 4] and rd.surviving(167) == [2, 3, 4]


def _random_fastq(path, rng, n_bc, stlfr=False):
    with open(path, "w") as f:
        idx = 0
        for b in range(n_bc):
            bc = "".join(rng.choice("ACGT") for _ in range(8)) if rng.random() > 0.15 else None
            for _ in range(rng.randint(1, 12)):
                idx += 1
                for mate in (1, 2):
                    s = "".join(rng.choice("ACGT" if rng.random() > 0.05 else "ACGTNacgtR")
                                for _ in range(rng.randint(25, 90)))
                    if stlfr:
                        h = f"@r{idx}#{b + 1}_{b + 2}_{b + 3}/{mate}" if bc else f"@r{idx}#0_0_0/{mate}"
                    else:
                        h = f"@r{idx} BX:Z:{bc}-1" if bc else f"@r{idx}"
                    f.write(f"{h}\n{s}\n+\n{'F' * len(s)}\n")


@pytest.mark.skipif(oracle.ref_tool("count_tnf") is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(6))
def test_live_against_reference_binaries(seed, tmp_path):
    """randomised inputs through the reference binaries themselves (only where oracle/_ref exists)"""
    rng = random.Random(seed)
    fq = str(tmp_path / "r.fq")
    _random_fastq(fq, rng, n_bc=rng.randint(3, 25), stlfr=seed % 3 == 2)
    mlen = rng.choice([0, 100, 300, 600])
    k_tnf = rng.choice([2, 3, 4, 5])
    k = rng.choice([5, 11, 15, 21, 27])
    w, v = rng.choice([(1, 6), (2, 9), (10, 400)])
    rd = oracle.Reads(fq)
    table = oracle.Table(k, threads=2).count(rd.all_seq())
    dump = str(tmp_path / "k.dump")
    table.dump(dump)
    out_t, out_a = str(tmp_path / "t.gz"), str(tmp_path / "a.gz")
    subprocess.run([oracle.ref_tool("count_tnf"), "-i", fq, "-k", str(k_tnf), "-l", str(mlen), "-t", "3", "-o", out_t],
                   check=True, stdout=subprocess.DEVNULL)
    subprocess.run([oracle.ref_tool("count_kmer"), "-i", fq, "-g", dump, "-k", str(k), "-w", str(w), "-v", str(v),
                    "-l", str(mlen), "-t", "3", "-o", out_a], check=True, stdout=subprocess.DEVNULL)
    names, tnf, abd = rd.features(mlen, k_tnf=k_tnf, k_abd=k, table=table, window=w, vsize=v, threads=2)
    with gzip.open(out_t, "rb") as f:
        assert _csv_bytes(tmp_path, names, tnf) == f.read()
    with gzip.open(out_a, "rb") as f:
        assert _csv_bytes(tmp_path, names, abd) == f.read()


def test_data_normalize_matches_reference_module():
    g = np.load(os.path.join(GOLDEN, "data_g4.npz"))
    abd, tnf, w = oracle.data_normalize(g["abd_in"], g["tnf_in"])
    assert abd.dtype == np.float32 and tnf.dtype == np.float32 and w.dtype == np.float64
    assert np.array_equal(abd, g["abd"]) and np.array_equal(tnf, g["tnf"]) and np.array_equal(w, g["weights"])
    assert np.array_equal(abd[3], g["item3_abd"]) and np.array_equal(tnf[3], g["item3_tnf"])


def test_vae_embedding_matches_reference_module():
    g = np.load(os.path.join(GOLDEN, "vae_g5.npz"))
    state = {k[len("state/"):]: g[k] for k in g.files if k.startswith("state/")}
    mu = oracle.vae_embedding(state, g["abd"], g["tnf"])
    scale = np.abs(g["mu"]).max()
    assert np.abs(mu - g["mu"]).max() <= 1e-6 * scale
