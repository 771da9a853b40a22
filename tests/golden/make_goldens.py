#!/usr/bin/env python3
"""Regenerate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference):
  * part A -- the reference's own ``count_tnf`` / ``count_kmer`` binaries, compiled by
    ``make -C oracle ref`` into oracle/_ref/ from /root/reference/src/cpptools, are run on the tiny
    FASTQ inputs written here; their CSV outputs are stored next to the inputs.  ``count_kmer`` needs a
    jellyfish dump; jellyfish is not installed, so the dump text is produced by the oracle's exact counter
    (the dump is an INPUT of the golden case and is stored as such).
  * part B -- ``/root/reference/src/data.py`` and ``src/models/VAENET.py`` are imported (two harness-side
    shims: a stub ``kneed`` module and ``np.Inf``) and evaluated on seeded inputs -> ``data_g4.npz``,
    ``vae_g5.npz``.
  * part C -- the reference's ``extract_reads`` binary (same build) on three inputs + clusters.tsv files ->
    ``bins_*/cluster_bin<label>.{fq,barcode}``.
  * part D -- the vendored ``third_parties/rph_kmeans`` library (its pure-python reducer; the Cython extension is
    not built) on seeded separable latents -> ``rph_g6.npz`` (labels, inertia).

Everything written is data (inputs + expected outputs); no reference source text is stored.
Usage:  python tests/golden/make_goldens.py
"""
import gzip
import json
import os
import random
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import oracle  # noqa: E402

REF = "/root/reference"


def rseq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def write_fastq(path, records, newline="\n"):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wt", newline="") as f:
        for rec in records:
            h, s = rec[0], rec[1]
            q = rec[2] if len(rec) > 2 else "I" * len(s)              # (an explicit quality line: the jellyfish-rule cases)
            f.write(f"{h}{newline}{s}{newline}+{newline}{q}{newline}")


def tenx_records(rng, plan, lo=40, hi=70, noisy=True):
    """plan: list of (barcode_or_None, n_pairs, style) -> interleaved records"""
    recs = []
    idx = 0
    for bc, n_pairs, style in plan:
        for _ in range(n_pairs):
            idx += 1
            pair = []
            for mate in (1, 2):
                s = list(rseq(rng, rng.randint(lo, hi)))
                if noisy:
                    r = rng.random()
                    if r < 0.15:
                        s[rng.randrange(len(s))] = "N"
                    elif r < 0.22:
                        j = rng.randrange(len(s)); s[j] = s[j].lower()
                    elif r < 0.27:
                        s[rng.randrange(len(s))] = rng.choice("RYKMSW")
                s = "".join(s)
                if bc is None:
                    h = f"@read{idx}/{mate}" if style == "slash" else f"@read{idx}"
                elif style == "nodash":
                    h = f"@read{idx} BX:Z:{bc}"
                elif style == "tab":
                    h = f"@read{idx}\tBX:Z:{bc}-1"
                elif style == "extra":
                    h = f"@read{idx} RG:Z:x BX:Z:{bc}-1 QT:Z:AAAA"
                elif style == "hashbx":
                    h = f"@read{idx}#9_9_9/{mate} BX:Z:{bc}-1"
                else:
                    h = f"@read{idx} BX:Z:{bc}-1"
                pair.append((h, s))
            recs.extend(pair)
    return recs


def stlfr_records(rng, plan, lo=40, hi=70):
    recs = []
    idx = 0
    for bc, n_pairs in plan:
        for _ in range(n_pairs):
            idx += 1
            for mate in (1, 2):
                recs.append((f"@V300_{idx}#{bc}/{mate}", rseq(rng, rng.randint(lo, hi))))
    return recs


def run(cmd):
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


def gunzip_to(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read())
    os.remove(src)


def part_a():
    ref_tnf, ref_kmer = oracle.ref_tool("count_tnf"), oracle.ref_tool("count_kmer")
    assert ref_tnf and ref_kmer, "run `make -C oracle ref` first"
    rng = random.Random(20211)
    cases = []
    inputs = {}

    # ---- inputs ------------------------------------------------------------------------------
    # 10x: tagged runs, a tag without '-', untagged pairs at start / middle / end (sorted-tail case),
    # a barcode that re-appears in a second non-adjacent run, a TAB-separated and an extra-tags header
    plan = [(None, 2, ""), ("AAACCCGG", 6, ""), ("AAACGTTA", 1, ""), ("AACGTTTC", 7, "nodash"),
            (None, 3, ""), ("ACGTACGT", 5, "tab"), ("AAACCCGG", 4, "extra"), ("CCGGTTAA", 9, ""),
            ("GGTTAACC", 2, ""), (None, 4, "")]
    write_fastq(os.path.join(HERE, "tenx_mixed.fq"), tenx_records(rng, plan))
    inputs["tenx_mixed"] = {"i": "tenx_mixed.fq"}
    # same content style, gzip'd, all clean bases, longer reads
    plan = [("AAAA", 8, ""), ("AAAC", 12, ""), ("AACC", 1, ""), ("ACCC", 10, ""), (None, 5, "")]
    write_fastq(os.path.join(HERE, "tenx_clean.fq.gz"), tenx_records(rng, plan, 100, 150, noisy=False))
    inputs["tenx_clean"] = {"i": "tenx_clean.fq.gz"}
    # header carrying both '#' and BX:Z (10x wins); first header untagged with '/1' (mode latches later)
    plan = [(None, 1, "slash"), ("TTTTAAAA", 5, "hashbx"), ("TTTTCCCC", 6, "hashbx")]
    write_fastq(os.path.join(HERE, "tenx_hashbx.fq"), tenx_records(rng, plan))
    inputs["tenx_hashbx"] = {"i": "tenx_hashbx.fq"}
    # stLFR raw headers incl. 0_0_0 (unbarcoded) in the middle and at the end
    plan = [("1_2_3", 5), ("1_2_4", 7), ("0_0_0", 3), ("7_8_9", 6), ("10_11_12", 1), ("0_0_0", 2)]
    write_fastq(os.path.join(HERE, "stlfr.fq"), stlfr_records(rng, plan))
    inputs["stlfr"] = {"i": "stlfr.fq"}
    # CRLF line ends: '\r' stays in every line (a non-base character, and part of the length test)
    plan = [("AAAA", 4, ""), ("CCCC", 5, ""), ("GGGG", 6, "")]
    write_fastq(os.path.join(HERE, "tenx_crlf.fq"), tenx_records(rng, plan, noisy=False), newline="\r\n")
    inputs["tenx_crlf"] = {"i": "tenx_crlf.fq"}
    # one barcode only; and a file whose last line has no newline
    write_fastq(os.path.join(HERE, "tenx_single.fq"), tenx_records(rng, [("ACACACAC", 9, "")]))
    with open(os.path.join(HERE, "tenx_single.fq"), "rb+") as f:
        f.seek(-1, 2); f.truncate()
    inputs["tenx_single"] = {"i": "tenx_single.fq"}
    # fixed read length 20 => run string length = 42 * pairs: exercise size()==mlen and mlen+-1
    plan = [("AAAAAAAA", 3, ""), ("CCCCCCCC", 4, ""), ("GGGGGGGG", 5, ""), ("TTTTTTTT", 6, "")]
    write_fastq(os.path.join(HERE, "tenx_len.fq"), tenx_records(rng, plan, 20, 20, noisy=False))
    inputs["tenx_len"] = {"i": "tenx_len.fq"}
    # %g: a count above 999999 prints in exponent form.  poly-A reads, gz keeps it tiny.
    recs = []
    for bc, n in (("AAAA", 1), ("CCCC", 31), ("GGGG", 3)):
        for j in range(n):
            for mate in (1, 2):
                recs.append((f"@p{bc}{j} BX:Z:{bc}-1", "A" * 20000))
    write_fastq(os.path.join(HERE, "polya.fq.gz"), recs)
    inputs["polya"] = {"i": "polya.fq.gz"}
    # paired files: identical names, one pair with different names, one with different barcodes
    r1, r2 = [], []
    idx = 0
    for bc, n in (("AAAA", 4), ("CCCC", 6), ("GGGG", 5)):
        for j in range(n):
            idx += 1
            h1 = h2 = f"@pr{idx} BX:Z:{bc}-1"
            if idx == 3:
                h2 = f"@pr{idx}x BX:Z:{bc}-1"
            if idx == 7:
                h2 = f"@pr{idx} BX:Z:TTTT-1"
            r1.append((h1, rseq(rng, rng.randint(40, 70))))
            r2.append((h2, rseq(rng, rng.randint(40, 70))))
    write_fastq(os.path.join(HERE, "pair_R1.fq"), r1)
    write_fastq(os.path.join(HERE, "pair_R2.fq"), r2)
    inputs["pair"] = {"1": "pair_R1.fq", "2": "pair_R2.fq"}
    # the two jellyfish rules the reference's own counters do not share.  Paired files with mixed quality
    # characters: in this branch jellyfish runs with --min-qual-char=? (feature.py:76-83) and reads a base below '?' as N;
    # reads repeat so that k-mers through a masked base still occur elsewhere with good quality.  And soft-masked reads:
    # jellyfish counts lower-case bases, count_tnf / count_kmer reset on them.
    genome = rseq(rng, 400)
    r1, r2 = [], []
    idx = 0
    for bc, n in (("AAAA", 5), ("CCCC", 7), ("GGGG", 6)):
        for j in range(n):
            idx += 1
            h1 = h2 = f"@pq{idx} BX:Z:{bc}-1"
            if idx == 4:
                h2 = f"@pq{idx}x BX:Z:{bc}-1"                         # skipped pair: its reads still reach jellyfish
            a, b = rng.randint(0, 300), rng.randint(0, 300)
            s1, s2 = genome[a:a + rng.randint(45, 80)], genome[b:b + rng.randint(45, 80)]
            q1 = "".join(rng.choice("I" * 12 + "?@>5#") for _ in s1)
            q2 = "".join(rng.choice("I" * 12 + "?@>5#") for _ in s2)
            if idx == 9:
                q1 = "#" * len(s1)                                      # a read without a single trusted base
            r1.append((h1, s1, q1))
            r2.append((h2, s2, q2))
    write_fastq(os.path.join(HERE, "pairq_R1.fq"), r1)
    write_fastq(os.path.join(HERE, "pairq_R2.fq"), r2)
    inputs["pairq"] = {"1": "pairq_R1.fq", "2": "pairq_R2.fq"}
    recs = []
    idx = 0
    for bc, n in (("ACAC", 4), ("GTGT", 6), ("TTAA", 5)):
        for j in range(n):
            idx += 1
            for mate in (1, 2):
                a = rng.randint(0, 300)
                sq = list(genome[a:a + rng.randint(50, 90)])
                for _ in range(rng.randint(0, 2)):                      # soft-masked stretches
                    x = rng.randint(0, len(sq) - 1)
                    for y in range(x, min(len(sq), x + rng.randint(1, 25))):
                        sq[y] = sq[y].lower()
                if rng.random() < 0.2:
                    sq[rng.randint(0, len(sq) - 1)] = "N"
                recs.append((f"@sm{idx} BX:Z:{bc}-1", "".join(sq)))
    write_fastq(os.path.join(HERE, "soft.fq"), recs)
    inputs["soft"] = {"i": "soft.fq"}

    def in_args(spec):
        out = []
        for flag, fn in spec.items():
            out += [f"-{flag}", os.path.join(HERE, fn)]
        return out

    # ---- count_tnf --------------------------------------------------------------------------
    tnf_jobs = [("tenx_mixed", 4, 100), ("tenx_mixed", 3, 0), ("tenx_mixed", 2, 300), ("tenx_clean", 4, 2000),
                ("tenx_clean", 5, 1000), ("tenx_hashbx", 4, 100), ("stlfr", 4, 100), ("stlfr", 3, 400),
                ("tenx_crlf", 4, 100), ("tenx_single", 4, 100), ("tenx_len", 4, 167), ("tenx_len", 4, 168),
                ("tenx_len", 4, 169), ("polya", 4, 1000), ("pair", 4, 100), ("pair", 1, 100),
                ("pairq", 4, 100), ("soft", 4, 100)]
    for name, k, mlen in tnf_jobs:
        out = f"{name}.tnf.k{k}.l{mlen}.csv"
        tmp = os.path.join(HERE, out + ".gz")
        run([ref_tnf] + in_args(inputs[name]) + ["-k", str(k), "-l", str(mlen), "-t", "2", "-o", tmp])
        gunzip_to(tmp, os.path.join(HERE, out))
        cases.append({"tool": "count_tnf", "input": inputs[name], "k": k, "min_len": mlen, "expect": out})

    # ---- count_kmer -------------------------------------------------------------------------
    # dump text = exact canonical counts of every read of the input (what jellyfish -C would report);
    # "holes" variants drop every 3rd dump line to exercise the absent-k-mer branch.
    kmer_jobs = [("tenx_mixed", 5, 1, 6, 100, False), ("tenx_mixed", 5, 10, 400, 100, False),
                 ("tenx_mixed", 15, 1, 6, 100, False), ("tenx_mixed", 21, 10, 400, 0, True),
                 ("tenx_clean", 15, 10, 400, 2000, False), ("tenx_clean", 21, 1, 6, 1000, False),
                 ("tenx_clean", 7, 2, 50, 1000, True), ("stlfr", 15, 1, 6, 100, False),
                 ("stlfr", 4, 3, 7, 100, False), ("tenx_crlf", 11, 1, 6, 100, False),
                 ("tenx_len", 9, 1, 6, 168, False), ("polya", 15, 10, 400, 1000, False),
                 ("polya", 3, 100000, 400, 1000, False), ("pair", 15, 1, 6, 100, False),
                 ("pair", 21, 10, 400, 100, True), ("tenx_hashbx", 31, 1, 6, 100, False),
                 # dumps under jellyfish's rules (quality threshold of the paired branch, lower-case bases count)
                 ("pairq", 15, 1, 6, 100, False), ("pairq", 21, 1, 6, 100, False), ("pairq", 9, 2, 50, 100, False),
                 ("soft", 15, 1, 6, 100, False), ("soft", 21, 1, 6, 100, False)]
    for name, k, w, v, mlen, holes in kmer_jobs:
        spec = inputs[name]
        rd = oracle.Reads(os.path.join(HERE, spec.get("i") or spec["1"]),
                          os.path.join(HERE, spec["2"]) if "2" in spec else None)
        jf = name in ("pairq", "soft")
        tab = oracle.Table(k).count(rd.all_seq(), lowercase_is_base=jf)
        dump = f"{name}.k{k}{'.holes' if holes else ''}.dump"
        dpath = os.path.join(HERE, dump)
        tab.dump(dpath)
        lines = sorted(open(dpath).read().splitlines())
        if holes:
            lines = [ln for i, ln in enumerate(lines) if i % 3 != 1]
        with open(dpath, "w") as f:
            f.write("\n".join(lines) + "\n")
        out = f"{name}.abd.k{k}.w{w}.v{v}.l{mlen}{'.holes' if holes else ''}.csv"
        tmp = os.path.join(HERE, out + ".gz")
        run([ref_kmer] + in_args(spec) + ["-g", dpath, "-k", str(k), "-w", str(w), "-v", str(v),
                                          "-l", str(mlen), "-t", "2", "-o", tmp])
        gunzip_to(tmp, os.path.join(HERE, out))
        cases.append({"tool": "count_kmer", "input": spec, "k": k, "window": w, "vsize": v, "min_len": mlen,
                      "dump": dump, "holes": holes, "expect": out, "jellyfish_rules": jf})

    # column-order anchors
    import hashlib
    anchors = {}
    for k in (2, 3, 4):
        hdr = ",".join(oracle.code_to_kmer(int(c), k) for c in oracle.tnf_columns(k))
        anchors[str(k)] = {"ncols": len(oracle.tnf_columns(k)), "sha256_16": hashlib.sha256(hdr.encode()).hexdigest()[:16]}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump({"cases": cases, "tnf_column_anchors": anchors}, f, indent=1)
    print(f"part A: {len(cases)} cases")


def part_b():
    sys.path[:0] = [os.path.join(REF, "src"), os.path.join(REF, "third_parties", "rph_kmeans")]
    kneed = types.ModuleType("kneed"); kneed.KneeLocator = object      # rph_kmeans/k_selection.py imports it
    sys.modules["kneed"] = kneed
    if not hasattr(np, "Inf"):
        np.Inf = np.inf                                                 # utils.py:30 (NumPy < 2 spelling)
    import torch
    from data import Data
    from models.VAENET import VAENET, VaritionalAutoEncoder

    # G4: Data on integer matrices, including an all-zero row and a single-hot row
    rs = np.random.RandomState(4)
    abd = rs.poisson(3.0, size=(37, 400)).astype(np.int64) * (rs.rand(37, 400) < 0.2)
    tnf = rs.poisson(40.0, size=(37, 136)).astype(np.int64)
    abd[5] = 0; tnf[9] = 0; abd[11] = 0; abd[11, 17] = 123456
    names = np.array([f"bc{i}" for i in range(37)], dtype=object)
    d = Data(names, abd, tnf)
    item = d[3]
    np.savez_compressed(os.path.join(HERE, "data_g4.npz"), abd_in=abd, tnf_in=tnf, abd=d.abd, tnf=d.tnf,
                        weights=d.weights, item3_abd=item["abd"], item3_tnf=item["tnf"])

    # G5: the reference network (reduced hidden sizes keep the fixture small; same code path), a few
    # reference training steps so BatchNorm running stats and all weights are non-trivial, then eval.
    torch.manual_seed(2021)
    vn = VAENET(abd_dim=400, tnf_dim=136, latent_size=32, num_classes=5, epochs=1, cuda=False, num_gpus=1,
                lr=0.005, dropout=0.2, alpha=0.1, w_kl=0.015, weight_decay=0.0001)
    vn.network = VaritionalAutoEncoder(400, 136, hidden_sizes=[48, 40], dropout=0.2)
    net = vn.network
    x_abd = torch.from_numpy(np.abs(rs.randn(96, 400)).astype(np.float32)); x_abd /= x_abd.sum(1, keepdim=True)
    x_tnf = torch.from_numpy(np.abs(rs.randn(96, 136)).astype(np.float32)); x_tnf /= x_tnf.sum(1, keepdim=True)
    opt = torch.optim.Adam(net.parameters(), lr=0.005, weight_decay=0.0001)
    net.train()
    for _ in range(5):
        opt.zero_grad()
        loss = vn.unlabeled_loss(net(x_abd, x_tnf))["total"]
        loss.backward()
        opt.step()
    net.eval()
    with torch.no_grad():
        mu = net.emebdding(x_abd[:64], x_tnf[:64]).numpy()
        torch.manual_seed(77)                      # eval mode: epsilon is the first draw after the seed
        out = net(x_abd[:64], x_tnf[:64])
        losses = vn.unlabeled_loss(out)
    torch.manual_seed(77)
    eps = torch.randn(64, 32).numpy()
    blob = {f"state/{k}": v.numpy() for k, v in net.state_dict().items()}
    blob.update(abd=x_abd[:64].numpy(), tnf=x_tnf[:64].numpy(), mu=mu, epsilon=eps,
                fwd_mu=out["mu"].numpy(), fwd_logsigma=out["logsigma"].numpy(),
                fwd_abd_rec=out["abd_rec"].numpy(), fwd_tnf_rec=out["tnf_rec"].numpy(),
                loss_total=np.float64(losses["total"].item()), loss_abd=np.float64(losses["abd_rec"].item()),
                loss_tnf=np.float64(losses["tnf_rec"].item()), loss_kl=np.float64(losses["kl_loss"].item()),
                wa=np.float64(vn.wa), wt=np.float64(vn.wt), w_kl=np.float64(vn.w_kl))
    np.savez_compressed(os.path.join(HERE, "vae_g5.npz"), **blob)
    print("part B: data_g4.npz vae_g5.npz")

    # G5b: the network as the reference builds it by default -- VaritionalAutoEncoder(400, 136): hidden_sizes [512, 512],
    # latent 32 (VAENET.py:193) --, seed 2021, three reference training steps on 128 rows, then eval: the embedding of 64
    # rows and one full forward with its loss terms.  The whole state is stored (4 MB): the weights of a trained net are
    # what they are.
    torch.manual_seed(2021)
    vn = VAENET(abd_dim=400, tnf_dim=136, latent_size=32, num_classes=30, epochs=1, cuda=False, num_gpus=1,
                lr=0.005, dropout=0.2, alpha=0.1, w_kl=0.015, weight_decay=0.0001)
    net = vn.network                                    # the constructor's own network: the default layer sizes
    assert [m.out_features for m in net.modules() if isinstance(m, torch.nn.Linear)][:2] == [512, 512]
    rs = np.random.RandomState(55)
    x_abd = torch.from_numpy(rs.poisson(2.0, size=(128, 400)).astype(np.float32) + 1e-3); x_abd /= x_abd.sum(1, keepdim=True)
    x_tnf = torch.from_numpy(rs.poisson(30.0, size=(128, 136)).astype(np.float32) + 1e-3); x_tnf /= x_tnf.sum(1, keepdim=True)
    opt = torch.optim.Adam(net.parameters(), lr=0.005, weight_decay=0.0001)
    net.train()
    for _ in range(3):
        opt.zero_grad()
        loss = vn.unlabeled_loss(net(x_abd, x_tnf))["total"]
        loss.backward()
        opt.step()
    net.eval()
    with torch.no_grad():
        mu = net.emebdding(x_abd[:64], x_tnf[:64]).numpy()
        torch.manual_seed(78)
        out = net(x_abd[:64], x_tnf[:64])
        losses = vn.unlabeled_loss(out)
    torch.manual_seed(78)
    eps = torch.randn(64, 32).numpy()
    blob = {f"state/{k}": v.numpy() for k, v in net.state_dict().items()}
    blob.update(abd=x_abd[:64].numpy(), tnf=x_tnf[:64].numpy(), mu=mu, epsilon=eps,
                fwd_mu=out["mu"].numpy(), fwd_logsigma=out["logsigma"].numpy(),
                fwd_abd_rec=out["abd_rec"].numpy(), fwd_tnf_rec=out["tnf_rec"].numpy(),
                loss_total=np.float64(losses["total"].item()), loss_abd=np.float64(losses["abd_rec"].item()),
                loss_tnf=np.float64(losses["tnf_rec"].item()), loss_kl=np.float64(losses["kl_loss"].item()),
                wa=np.float64(vn.wa), wt=np.float64(vn.wt), w_kl=np.float64(vn.w_kl))
    np.savez_compressed(os.path.join(HERE, "vae_g5b_512.npz"), **blob)
    print("part B: vae_g5b_512.npz")


def part_c():
    """bin writer: the reference's extract_reads binary on three inputs -> every file it writes"""
    ref = oracle.ref_tool("extract_reads")
    assert ref, "run `make -C oracle ref` first"
    import shutil
    import tempfile
    jobs = [("bins_tenx", {"i": "tenx_mixed.fq"}, "3\tAAACCCGG,ACGTACGT\n-1\tAACGTTTC\n0\tCCGGTTAA,GGTTAACC,AAACCCGG\n7\t\n"),
            ("bins_stlfr", {"i": "stlfr.fq"}, "1\t1_2_3,7_8_9\n2\t10_11_12\n"),
            ("bins_pair", {"1": "pair_R1.fq", "2": "pair_R2.fq"}, "5\tAAAA\n6\tCCCC,GGGG\n")]
    man = []
    for name, spec, tsv in jobs:
        out = os.path.join(HERE, name)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        with open(os.path.join(out, "clusters.tsv"), "w") as f:
            f.write(tsv)
        args = []
        for flag, fn in spec.items():
            args += [f"-{flag}", os.path.join(HERE, fn)]
        run([ref] + args + ["-c", os.path.join(out, "clusters.tsv"), "-o", os.path.join(out, "cluster")])
        man.append({"dir": name, "input": spec, "files": sorted(f for f in os.listdir(out) if f.startswith("cluster_"))})
    with open(os.path.join(HERE, "manifest.json")) as f:
        m = json.load(f)
    m["bin_writer"] = man
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(m, f, indent=1)
    print(f"part C: {len(man)} bin-writer cases")


def part_d():
    """rph_kmeans (vendored library, python reducer: the Cython extension is not built) on separable latents:
    labels are not bit-reproducible across hash-map orders, so the fixture pins inertia and the partition"""
    sys.path[:0] = [os.path.join(REF, "third_parties", "rph_kmeans")]
    kneed = types.ModuleType("kneed"); kneed.KneeLocator = object
    sys.modules["kneed"] = kneed
    import warnings
    from rph_kmeans import RPHKMeans
    rs = np.random.RandomState(6)
    sizes = [700, 300, 150, 60, 40, 350]
    centers = rs.randn(len(sizes), 32) * 0.6
    X = np.concatenate([c + 0.05 * rs.randn(n, 32) for c, n in zip(centers, sizes)]).astype(np.float32)
    truth = np.concatenate([np.full(n, i) for i, n in enumerate(sizes)])
    perm = rs.permutation(len(X))
    X, truth = X[perm], truth[perm]
    np.random.seed(2021)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clt = RPHKMeans(n_init=20, n_clusters=len(sizes), verbose=0, max_point=200)
        labels = clt.fit_predict(X)
    np.savez_compressed(os.path.join(HERE, "rph_g6.npz"), X=X, truth=truth, labels=labels.astype(np.int32),
                        inertia=np.float64(clt.inertia_), n_clusters=np.int32(len(sizes)), max_point=np.int32(200))
    print("part D: rph_g6.npz inertia", clt.inertia_)


if __name__ == "__main__":
    parts = sys.argv[1:] or ["a", "b", "c", "d"]
    for part in parts:
        {"a": part_a, "b": part_b, "c": part_c, "d": part_d}[part]()
