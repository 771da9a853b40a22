"""Steps 1-3 of the orchestrator end to end on SURVEY 8d's C1 stand-in for BASELINE config 1 (the example data of the
reference are git-LFS pointers): a 100 k-pair 10x-style synthetic FASTQ.gz, ``-c 10``, checking the file layout the
reassembly stage consumes."""
import gzip
import shutil
import os

import numpy as np
import pytest
import torch

from pangaea_amd import synth

pytestmark = pytest.mark.gpu


def test_steps_1_to_3_write_the_reference_layout(tmp_path):
    from pangaea_amd import pangaea
    cfg = synth.SynthConfig(n_pairs=100_000, n_barcodes=490, n_genomes=10, genome_len=200_000, fragment=40_000, seed=3)
    s = synth.generate(cfg)
    plain = str(tmp_path / "reads.sorted.fastq")
    synth.write_fastq(s, cfg, plain)
    fq = plain + ".gz"
    with open(plain, "rb") as src, gzip.open(fq, "wb", compresslevel=1) as dst:
        shutil.copyfileobj(src, dst, 1 << 22)
    os.remove(plain)
    out = str(tmp_path / "out")
    pangaea.main(["-i", fq, "-o", out, "-c", "10", "-k", "21", "-l", "2000", "-e", "3", "-b", "64", "-st", "1,2,3", "-t", "8"])
    for rel in ("1.features/feature_finished", "1.features/tnf.m2000.gz", "1.features/tnf.m2000.pkl",
                "1.features/abundance.k21.v400.w10.m2000.gz", "1.features/abundance.k21.v400.w10.m2000.pkl",
                "2.vae/train_model.pk", "2.vae/latent.npz", "2.vae/barcodes.npz", "2.vae/model_finished",
                "3.clustering/clusters.npz", "3.clustering/clusters.tsv", "3.clustering/clustering_finished", "log"):
        assert os.path.isfile(os.path.join(out, rel)), rel
    latent = np.load(os.path.join(out, "2.vae/latent.npz"))["arr_0"]
    barcodes = np.load(os.path.join(out, "2.vae/barcodes.npz"))["arr_0"]
    assert latent.shape == (490, 32) and latent.dtype == np.float32 and len(barcodes) == 490
    state = torch.load(os.path.join(out, "2.vae/train_model.pk"), map_location="cpu")
    assert {"encoder.0.weight", "encoder.1.running_mean", "encoder.4.weight", "encoder.5.running_var", "l_mu.weight",
            "l_sigma.bias", "decoder.0.weight", "decoder.5.weight", "output.weight"} <= set(state)
    labels = np.load(os.path.join(out, "3.clustering/clusters.npz"))["arr_0"]
    assert labels.shape == (490,) and set(labels) <= set(range(10)) and len(set(labels)) > 1
    bins = sorted(f for f in os.listdir(os.path.join(out, "3.clustering")) if f.endswith(".fq"))
    assert bins == [f"cluster_bin{c}.fq" for c in sorted(set(labels))]
    # every barcoded pair lands in exactly one bin (4 lines per read, 2 reads per pair)
    lines = sum(sum(1 for _ in open(os.path.join(out, "3.clustering", b))) for b in bins)
    assert lines == 8 * cfg.pairs_per_barcode * cfg.n_barcodes
    # resume: everything is finished, a second call touches nothing
    before = os.path.getmtime(os.path.join(out, "2.vae/latent.npz"))
    pangaea.main(["-i", fq, "-o", out, "-c", "10", "-k", "21", "-st", "1,2,3"])
    assert os.path.getmtime(os.path.join(out, "2.vae/latent.npz")) == before
