"""BASELINE.json's configurations at (one GPU's share of) their real sizes, through the paths that ship.

config 2   10 M pairs / 50 k barcodes / k = 21: the fused count + lookup paths (super-k-mer pipeline, key-partitioned
           pipeline) == the lookup kernel, table total and row sums account for every k-mer, spot rows == the oracle
           (abundance rows too: the oracle counts the spot rows' k-mers over the whole text)
mid scale  2 M pairs: the WHOLE table and 25+ rows of both matrices == the oracle, for both fused pipelines
config 3   one GPU's share (25 M pairs / 125 k barcodes): more than 65 536 rows -> the two-pass row shuffle
config 4   stLFR headers, 1 M pairs written as FASTQ and taken through ``Feature`` (ingest, header grammar, caches)
config 5   hybrid mode: long-read names as barcodes, Poisson(100) pairs each, encode + ``clustering_rph_kmeans(-c 40)``:
           8 M pairs (78 k ragged rows: the count kernel's one-pass row scatter with 2048 digits) and one GPU's share of the
           real configuration, 25 M pairs (~250 k ragged rows: 4-byte words, two-pass row shuffle, the (row, slot) word
           within 5 % of its 2^18-row limit)
Integer matrices bit for bit (count_tnf.cpp:78-113, count_kmer.cpp:55-108, jellyfish count -C of feature.py:94).
"""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from pangaea_amd import _lib, kmer, synth
from pangaea_amd.reads import Rows

from .conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _text(s, chunk=1 << 25) -> np.ndarray:
    """the stream as characters, decoded piecewise (a 10 M-pair stream is 3 G characters)"""
    out = np.empty(s.n_chars, dtype=np.uint8)
    for a in range(0, s.n_chars, chunk):
        b = min(s.n_chars, a + chunk)
        out[a:b] = np.frombuffer(s.decode(a, b), dtype=np.uint8)
    return out


def _kmer_ends(valid_words: torch.Tensor, k: int) -> int:
    """number of positions that end a run of >= k valid characters (torch restatement of the kernels' validity rule)"""
    v = valid_words.to(torch.int64) & 0xFFFFFFFF
    prev = torch.cat([v.new_zeros(1), v[:-1]])
    r = (v << 32) | prev
    ln = 1
    while 2 * ln <= k:
        r = r & (r << ln)
        ln *= 2
    if ln < k:
        r = r & (r << (k - ln))
    hi = (r >> 32) & 0xFFFFFFFF
    total = 0
    for b in range(32):
        total += int(((hi >> b) & 1).sum().item())
    return total


def _spot_rows_equal_oracle(s, rows, text, tnf, abd, pick, k=21, window=10, vsize=400, threads=16):
    seqs = [text[int(rows.start[r]):int(rows.end[r])].tobytes() for r in pick]
    known = oracle.Table.zeroed_keys_of(k, seqs, threads=threads).count_known(text)     # exact global counts of the rows' k-mers
    for r, seq in zip(pick, seqs):
        assert np.array_equal(tnf[r].cpu().numpy(), oracle.tnf_row(seq, 4)), f"TNF row {r}"
        assert np.array_equal(abd[r].cpu().numpy(), oracle.abd_row(seq, k, known, window, vsize)), f"abundance row {r}"


def test_config2_full_size_fused_paths():
    cfg = synth.SynthConfig(n_pairs=10_000_000, n_barcodes=50_000)
    s = synth.generate(cfg, device=DEV, chunk_pairs=1 << 17, with_names=False)
    rows = s.rows(2000)
    assert len(rows) == 50_000
    plan = kmer.Plan(rows, DEV)
    # (a) what Feature and bench.py run on one GPU: count_kmers picks the super-k-mer pipeline, lookups inside the count
    mini = kmer.count_kmers(s, 21, rows=plan, emit=(10, 400))
    assert mini.kind == "mini" and mini._emitted == (10, 400)
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=mini, window=10, vsize=400)
    # (b) the key-partitioned pipeline, fused the same way
    fused = kmer.KmerTable.with_slots(21, DEV, 29, 14).count(s, rows=plan, emit=(10, 400))
    assert fused._emitted == (10, 400)
    _, abd_f = kmer.features(s, plan, k_tnf=None, table=fused, window=10, vsize=400)
    assert torch.equal(abd_f, abd)
    # (c) the lookup kernel on either table
    tnf_l, abd_l = kmer.features(s, rows, k_tnf=4, table=mini, window=10, vsize=400)
    assert torch.equal(tnf_l, tnf) and torch.equal(abd_l, abd)
    _, abd_l2 = kmer.features(s, rows, k_tnf=None, table=fused, window=10, vsize=400)
    assert torch.equal(abd_l2, abd)
    torch.cuda.synchronize()
    # every valid 21-mer occurrence is in the table (either form), every occurrence inside a row is binned or >= 4000
    n21 = _kmer_ends(s.valid, 21)
    for t in (mini, fused):
        assert int((t.compact() & ((1 << 22) - 1)).sum().item()) == n21
    _, abd_all = kmer.features(s, rows, k_tnf=None, table=mini, window=1 << 11, vsize=1 << 10)
    assert (abd.to(torch.int64).sum(1) <= abd_all.to(torch.int64).sum(1)).all()
    cpp = cfg.chars_per_pair
    comp = Rows(np.arange(2), ["head", "tail"], np.array([0, int(rows.end[-1])]), np.array([cpp, s.n_chars]))
    _, abd_c = kmer.features(s, comp, k_tnf=None, table=mini, window=1 << 11, vsize=1 << 10)
    assert int(abd_all.to(torch.int64).sum().item()) + int(abd_c.sum().item()) == n21
    del fused, abd_f, abd_l, abd_l2, abd_all
    # spot rows against the oracle: TNF and abundance
    text = _text(s)
    _spot_rows_equal_oracle(s, rows, text, tnf, abd, [0, 17, 24_999, 31_337, 49_999])


@pytest.mark.parametrize("kind", ["mini", "hash"])
def test_mid_scale_fused_table_and_rows_against_oracle(kind):
    cfg = synth.SynthConfig(n_pairs=2_000_000, n_barcodes=10_000, n_genomes=16, genome_len=1_000_000, seed=91)
    s = synth.generate(cfg, device=DEV, chunk_pairs=1 << 17, with_names=False)
    rows = s.rows(2000)
    plan = kmer.Plan(rows, DEV)
    t = (kmer.KmerTable.mini_with_slots(21, DEV, 28, 14) if kind == "mini" else kmer.KmerTable.with_slots(21, DEV, 28, 14))
    t.count(s, rows=plan, emit=(10, 400))
    assert t._emitted == (10, 400)
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=10, vsize=400)
    text = _text(s)
    otab = oracle.Table(21, threads=16).count(text)
    assert all(np.array_equal(x, y) for x, y in zip(t.items(), otab.items()))
    for r in range(0, len(rows), len(rows) // 30):
        seq = text[int(rows.start[r]):int(rows.end[r])].tobytes()
        assert np.array_equal(tnf[r].cpu().numpy(), oracle.tnf_row(seq, 4))
        assert np.array_equal(abd[r].cpu().numpy(), oracle.abd_row(seq, 21, otab, 10, 400))


def test_config3_one_gpu_share_two_pass_row_shuffle():
    """25 M pairs / 125 k barcodes (200 M pairs / 1 M barcodes over 8 GPUs): more rows than one 1024-way pass can
    take.  Key-partitioned pipeline as count_kmers sizes it, and the super-k-mer pipeline at its largest geometry."""
    cfg = synth.SynthConfig(n_pairs=25_000_000, n_barcodes=125_000, seed=2023)
    s = synth.generate(cfg, device=DEV, chunk_pairs=1 << 17, with_names=False)
    rows = s.rows(2000)
    assert len(rows) == 125_000
    plan = kmer.Plan(rows, DEV)
    t = kmer.count_kmers(s, 21, rows=plan, emit=(10, 400))
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=10, vsize=400)
    n21 = _kmer_ends(s.valid, 21)
    assert int((t.compact() & ((1 << 22) - 1)).sum().item()) == n21
    t.release_workspaces()                 # (two tables' scratch -- 150 GB each at this size -- need not coexist)
    m = kmer.KmerTable.mini_with_slots(21, DEV, 29, 14).count(s, rows=plan, emit=(10, 400))
    _, abd_m = kmer.features(s, plan, k_tnf=None, table=m, window=10, vsize=400)
    assert torch.equal(abd_m, abd)
    # (sorting 2 x 285 M entries to compare them one by one is the mid-scale test's job; here: as many entries, the same total)
    assert int(torch.count_nonzero(m.data).item()) == int(torch.count_nonzero(t.data).item())
    assert int((m.compact() & ((1 << 22) - 1)).sum().item()) == n21
    del m, abd_m
    # idempotence, and the rows of a slice by the lookup kernel
    t.reset().count(s, rows=plan, emit=(10, 400))
    _, abd2 = kmer.features(s, plan, k_tnf=None, table=t, window=10, vsize=400)
    assert torch.equal(abd2, abd)
    pick = np.arange(70_000, 70_400)
    sub = Rows(pick, [rows.names[i] for i in pick], rows.start[pick], rows.end[pick])
    tnf_s, abd_s = kmer.features(s, sub, k_tnf=4, table=t, window=10, vsize=400)
    assert torch.equal(abd_s, abd[70_000:70_400]) and torch.equal(tnf_s, tnf[70_000:70_400])
    for r in (0, 65_536, 124_999):
        seq = s.decode(int(rows.start[r]), int(rows.end[r]))
        assert np.array_equal(tnf[r].cpu().numpy(), oracle.tnf_row(seq, 4))


def test_fifty_million_pairs_on_one_gpu_counted_in_pieces(monkeypatch, capsys):
    """more pairs than one GPU's share of any BASELINE configuration: 50 M pairs / 250 k barcodes on ONE GPU.  The stream still
    fits in one piece here (about 3.4 KB of scratch per pair), so both forms run: one count, and the same stream counted in two
    word ranges into the same table (``PANGAEA_MINI_PIECE_WORDS``; beyond about 70 M pairs ``KmerTable`` does that by itself).
    Same rows either way, table total = the valid 21-mers of the stream, spot rows against the oracle's exact counts; the pieces'
    rate stays within a quarter of the one-piece rate (the two printed; DESIGN.md has the measured figures)."""
    import time
    t_start = time.perf_counter()
    cfg = synth.SynthConfig(n_pairs=50_000_000, n_barcodes=250_000, seed=4242)
    s = synth.generate(cfg, device=DEV, chunk_pairs=1 << 17, with_names=False)
    torch.cuda.synchronize(); t_gen = time.perf_counter()
    rows = s.rows(2000)
    assert len(rows) == 250_000
    plan = kmer.Plan(rows, DEV)
    n21 = _kmer_ends(s.valid, 21)

    def timed(table_maker):
        # (twice: the first pass allocates ~170 GB through hipMalloc, seconds of it; the second finds the blocks in torch's cache)
        for again in (False, True):
            t = tnf = abd = None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            t = table_maker()
            tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=10, vsize=400)
            torch.cuda.synchronize()
            sec = time.perf_counter() - t0
        return t, tnf, abd, sec

    one, tnf, abd, sec_one = timed(lambda: kmer.count_kmers(s, 21, rows=plan, emit=(10, 400)))
    assert one.kind == "mini" and one._mini_pieces == 1
    geometry = (one.log2_slots, one.log2_bucket)
    assert int((one.compact() & ((1 << 22) - 1)).sum().item()) == n21
    peak_one = torch.cuda.max_memory_allocated(DEV)
    one.release_workspaces()
    del one
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats(DEV)
    # (first call of the pieces' kernels on a small stream: their code objects load here, not inside the timed count)
    small = synth.generate(synth.SynthConfig(n_pairs=40_000, n_barcodes=200, seed=1), device=DEV)
    monkeypatch.setenv("PANGAEA_MINI_PIECE_WORDS", str(256 * 256))
    warm = kmer.KmerTable.mini_with_slots(21, DEV, 24, 12).count(small, rows=kmer.Plan(small.rows(2000), DEV), emit=(10, 400))
    assert warm._mini_pieces > 1
    del warm, small
    monkeypatch.setenv("PANGAEA_MINI_PIECE_WORDS", str(s.n_words // 2 // 256 * 256 + 256))
    t, tnf_p, abd_p, sec_pieces = timed(lambda: kmer.KmerTable.mini_with_slots(21, DEV, *geometry).count(s, rows=plan, emit=(10, 400)))
    assert t.kind == "mini" and t._mini_pieces == 2
    assert torch.equal(abd_p, abd) and torch.equal(tnf_p, tnf)
    assert int((t.compact() & ((1 << 22) - 1)).sum().item()) == n21
    peak_pieces = torch.cuda.max_memory_allocated(DEV)
    with capsys.disabled():
        print(f"\n50 M pairs on one GPU: one piece {sec_one:.3f} s (peak {peak_one / 2**30:.0f} GiB), two pieces {sec_pieces:.3f} s "
              f"(peak {peak_pieces / 2**30:.0f} GiB): {50 / sec_one:.0f} vs {50 / sec_pieces:.0f} M pairs/s")
    assert sec_pieces <= 1.25 * sec_one
    t_gpu = time.perf_counter()
    text = _text(s)
    t_text = time.perf_counter()
    _spot_rows_equal_oracle(s, rows, text, tnf_p, abd_p, [0, 131_072, 249_999])
    with capsys.disabled():
        print(f"(test phases: generate {t_gen - t_start:.0f} s, counts {t_gpu - t_gen:.0f} s, decode {t_text - t_gpu:.0f} s, oracle {time.perf_counter() - t_text:.0f} s)")


def _args(tmp_path, **kw):
    d = dict(reads1="", reads2="", interleaved_reads="", output=str(tmp_path / "out"), min_length=2000, kmer=21,
             tnf_kmer=4, window_size=10, vector_size=400, threads=8)
    d.update(kw)
    os.makedirs(d["output"], exist_ok=True)
    return argparse.Namespace(**d)


def test_config4_stlfr_fastq_through_feature(tmp_path):
    """raw stLFR headers (@name#b1_b2_b3/1, 0_0_0 = no barcode; count_tnf.cpp:35-42), 1 M pairs from a FASTQ file through
    ``Feature``: every row of both matrices and every name == the oracle's reading of the same file"""
    from pangaea_amd.feature import Feature
    cfg = synth.SynthConfig(n_pairs=1_000_000, n_barcodes=5_000, n_genomes=16, genome_len=1_000_000, seed=44)
    s = synth.generate(cfg, device=DEV, chunk_pairs=1 << 17)
    fq = str(tmp_path / "stlfr.fq")
    synth.write_fastq(s, cfg, fq, style="stlfr")
    args = _args(tmp_path, interleaved_reads=fq)
    names, abd, tnf = Feature(args, ROOT).extract_features()
    rd = oracle.Reads(fq)
    assert rd.mode == "stLFR"
    table = oracle.Table(21, threads=16).count(rd.all_seq())
    onames, otnf, oabd = rd.features(2000, k_tnf=4, k_abd=21, table=table, window=10, vsize=400, threads=16)
    assert list(names) == onames and len(onames) == 5_000 and names[0] == "1_1_1"
    assert np.array_equal(tnf, otnf) and np.array_equal(abd, oabd)
    for fn in ("tnf.m2000.gz", "tnf.m2000.pkl", "abundance.k21.v400.w10.m2000.gz", "abundance.k21.v400.w10.m2000.pkl", "feature_finished"):
        assert os.path.isfile(os.path.join(args.output, "1.features", fn)), fn


def test_config5_hybrid_rows_encode_and_rph_kmeans():
    """hybrid mode: a long read's name is the barcode of the short pairs mapped to it (assign_barcodes.cpp:156), Poisson(100)
    pairs each: 8 M pairs -> ~78 k rows (two-pass row shuffle), L1-normalise, VAE encode, RPH-KMeans with -c 40"""
    from pangaea_amd.clustering import clustering_rph_kmeans
    from pangaea_amd.data import Data
    from pangaea_amd.models.VAENET import VAENET
    cfg = synth.SynthConfig(n_pairs=8_000_000, n_barcodes=1, poisson_mean=100.0, seed=77)
    s = synth.generate(cfg, device=DEV, chunk_pairs=1 << 17, with_names=False)
    rows = s.rows(2000)
    n_rows = len(rows)
    assert n_rows > 70_000 and len(set(np.diff(rows.start).tolist())) > 50          # ragged barcodes
    plan = kmer.Plan(rows, DEV)
    t = kmer.count_kmers(s, 21, rows=plan, emit=(10, 400))
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=10, vsize=400)
    assert int((t.compact() & ((1 << 22) - 1)).sum().item()) == _kmer_ends(s.valid, 21)
    tnf_l, abd_l = kmer.features(s, rows, k_tnf=4, table=t, window=10, vsize=400)
    assert torch.equal(tnf_l, tnf) and torch.equal(abd_l, abd)
    text_rows = [0, 1, n_rows // 2, n_rows - 1]
    for r in text_rows:
        assert np.array_equal(tnf[r].cpu().numpy(), oracle.tnf_row(s.decode(int(rows.start[r]), int(rows.end[r])), 4))
    # rows -> Data -> encode -> clusters (random-init weights: shapes, finiteness and a full labelling are what is checked)
    torch.manual_seed(2021)
    d = Data(np.array(rows.names, dtype=object), abd, tnf, device=torch.device(DEV))
    oa, ot, ow = oracle.data_normalize(abd[:512].cpu().numpy(), tnf[:512].cpu().numpy())
    assert np.array_equal(d.abd[:512], oa) and np.array_equal(d.tnf[:512], ot) and np.array_equal(d.weights[:512], ow)
    vae = VAENET(400, 136, 32, 40, 1, True, 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
    vae.network.eval()
    mu = vae.encode(d)
    assert tuple(mu.shape) == (n_rows, 32) and bool(torch.isfinite(mu).all())
    labels = clustering_rph_kmeans(mu.cpu().numpy(), 40)
    assert labels.shape == (n_rows,) and 1 < len(np.unique(labels)) <= 40 and labels.min() >= 0


def test_config5_one_gpu_share_hybrid_rows_encode_and_rph_kmeans():
    """one GPU's share of BASELINE config 5 (200 M short pairs + 2 M long-read barcodes over 8 GPUs): 25 M pairs, Poisson(100)
    pairs per long read -> ~250 k ragged rows.  That is past the 131 072 rows the count kernel scatters in one pass (so: 4-byte
    (row, bin) words and the row shuffle's second pass) and just inside the 2^18 - 1 rows a (row, slot) provisional word can
    name with 2^14-slot buckets.  The shipped path (count_kmers with emit), table total = valid 21-mers, shuffle == lookup rows
    on a slice, TNF and abundance spot rows == the oracle (exact global counts of the rows' k-mers), then Data -> encode ->
    RPH-KMeans -c 40 (assign_barcodes.cpp:156, clustering.py:14-19, count_kmer.cpp:55-108)."""
    from pangaea_amd.clustering import clustering_rph_kmeans
    from pangaea_amd.data import Data
    from pangaea_amd.models.VAENET import VAENET
    cfg = synth.SynthConfig(n_pairs=25_000_000, n_barcodes=1, poisson_mean=100.0, seed=78)
    s = synth.generate(cfg, device=DEV, chunk_pairs=1 << 17, with_names=False)
    rows = s.rows(2000)
    n_rows = len(rows)
    assert 131_072 < n_rows < (1 << 18) - 1 and len(set(np.diff(rows.start[:5000]).tolist())) > 50       # ragged, two-pass, slot form
    plan = kmer.Plan(rows, DEV)
    t = kmer.count_kmers(s, 21, rows=plan, emit=(10, 400))
    assert t.kind == "mini" and t._emitted == (10, 400)
    tnf, abd = kmer.features(s, plan, k_tnf=4, table=t, window=10, vsize=400)
    assert int((t.compact() & ((1 << 22) - 1)).sum().item()) == _kmer_ends(s.valid, 21)
    for lo in (0, 131_000, n_rows - 400):                          # either side of the one-pass limit, and the last rows
        pick = np.arange(lo, lo + 400)
        sub = Rows(pick, [rows.names[i] for i in pick], rows.start[pick], rows.end[pick])
        tnf_s, abd_s = kmer.features(s, sub, k_tnf=4, table=t, window=10, vsize=400)
        assert torch.equal(abd_s, abd[lo:lo + 400]) and torch.equal(tnf_s, tnf[lo:lo + 400])
    text = _text(s)
    _spot_rows_equal_oracle(s, rows, text, tnf, abd, [0, 65_535, 131_071, 131_072, n_rows // 2 + 1, n_rows - 1])
    del text
    torch.manual_seed(2021)
    d = Data(np.array(rows.names, dtype=object), abd, tnf, device=torch.device(DEV))
    oa, ot, ow = oracle.data_normalize(abd[-512:].cpu().numpy(), tnf[-512:].cpu().numpy())
    assert np.array_equal(d.abd[-512:], oa) and np.array_equal(d.tnf[-512:], ot) and np.array_equal(d.weights[-512:], ow)
    vae = VAENET(400, 136, 32, 40, 1, True, 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
    vae.network.eval()
    mu = vae.encode(d)
    assert tuple(mu.shape) == (n_rows, 32) and bool(torch.isfinite(mu).all())
    labels = clustering_rph_kmeans(mu.cpu().numpy(), 40)
    assert labels.shape == (n_rows,) and 1 < len(np.unique(labels)) <= 40 and labels.min() >= 0
