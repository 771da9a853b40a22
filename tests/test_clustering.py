"""RPH-KMeans on torch against the vendored library's result (tests/golden/rph_g6.npz).  Bucket numbering and sklearn
internals make labels non-reproducible bit for bit: parity = same partition (ARI) and no worse inertia."""
import os

import numpy as np
import pytest
import torch
from sklearn.cluster import KMeans
from sklearn.metrics import adjusted_rand_score

from pangaea_amd import clustering

from .conftest import GOLDEN


def _fixture():
    g = np.load(os.path.join(GOLDEN, "rph_g6.npz"))
    return g["X"], g["truth"], g["labels"], float(g["inertia"]), int(g["n_clusters"]), int(g["max_point"])


def _check(device):
    X, truth, ref_labels, ref_inertia, k, max_point = _fixture()
    np.random.seed(2021)
    clt = clustering.RPHKMeans(n_init=20, n_clusters=k, max_point=max_point, device=device)
    labels = clt.fit_predict(X)
    assert labels.shape == (len(X),) and clt.cluster_centers_.shape == (k, 32)
    assert adjusted_rand_score(ref_labels, labels) >= 0.99
    assert adjusted_rand_score(truth, labels) >= 0.99
    assert clt.inertia_ <= 1.02 * ref_inertia
    assert clt.reduced_X_.shape[0] <= max_point and clt.rp_iter_ >= 1
    assert np.isclose(clt.reduced_X_weight_.sum(), len(X))
    # the reducer's labels point every sample at its reduced point
    assert clt.rp_labels_.max() < clt.reduced_X_.shape[0]
    assert np.array_equal(clt.predict(X), labels)


def test_rph_kmeans_cpu_torch_matches_reference_partition():
    _check("cpu")


@pytest.mark.gpu
def test_rph_kmeans_gpu_matches_reference_partition():
    _check("cuda:0")
    X, _, _, _, k, _ = _fixture()
    np.random.seed(1)
    labels = clustering.clustering_rph_kmeans(X, k)            # the reference's call: n_init=20, defaults
    assert len(np.unique(labels)) == k


def _lloyd_vs_sklearn(device):
    rs = np.random.RandomState(0)
    X = np.concatenate([rs.randn(200, 32) * 0.3 + c for c in rs.randn(5, 32) * 2]).astype(np.float32)
    init = X[rs.choice(len(X), 5, replace=False)]
    sk = KMeans(n_clusters=5, init=init, n_init=1).fit(X)
    labels, centers, inertia, _ = clustering.lloyd(torch.from_numpy(X).to(device), torch.from_numpy(init).to(device))
    assert adjusted_rand_score(sk.labels_, labels.cpu().numpy()) == 1.0
    assert np.allclose(centers.cpu().numpy(), sk.cluster_centers_, atol=1e-4)
    assert abs(inertia - sk.inertia_) <= 1e-3 * sk.inertia_


def test_lloyd_distance_step_equals_sklearn():
    """same start centres -> same fixed point as sklearn's Lloyd (the reference's final step, rph_kmeans_.py:154-155)"""
    _lloyd_vs_sklearn("cpu")


@pytest.mark.gpu
def test_lloyd_distance_step_equals_sklearn_on_the_gpu():
    _lloyd_vs_sklearn("cuda:0")


def _skeleton_vs_sklearn(device):
    """the weighted skeleton k-means (k-means++ seeding + weighted Lloyd on the device) against sklearn's weighted KMeans --
    what the reference calls on the reduced points (rph_kmeans_.py:121-124): same partition, inertia within 2 %"""
    rs = np.random.RandomState(5)
    cen = rs.randn(12, 32) * 3
    X = np.concatenate([rs.randn(150, 32) * 0.4 + c for c in cen]).astype(np.float32)
    w = rs.randint(1, 60, size=len(X)).astype(np.float32)
    sk = KMeans(n_clusters=12, n_init=10, random_state=0).fit(X, sample_weight=w)
    np.random.seed(3)
    centers, labels, inertia = clustering.weighted_kmeans(torch.from_numpy(X).to(device), torch.from_numpy(w).to(device), 12, n_init=3)
    assert tuple(centers.shape) == (12, 32) and adjusted_rand_score(sk.labels_, labels.cpu().numpy()) >= 0.99
    assert inertia <= 1.02 * sk.inertia_
    # seeding alone already lands one centre per blob most of the time, and never on a zero-weight point
    w0 = w.copy(); w0[:150] = 0
    gen = torch.Generator(device=device); gen.manual_seed(1)
    c0 = clustering.kmeans_plusplus(torch.from_numpy(X).to(device), torch.from_numpy(w0).to(device), 11, gen).cpu().numpy()
    assert not any((np.abs(X[:150] - c).sum(1) == 0).any() for c in c0)


def test_skeleton_kmeans_equals_sklearn():
    _skeleton_vs_sklearn("cpu")


@pytest.mark.gpu
def test_skeleton_kmeans_equals_sklearn_on_the_gpu():
    _skeleton_vs_sklearn("cuda:0")


def test_cluster_barcode_reads_writes_the_bin_layout(tmp_path):
    import argparse
    X, _, _, _, k, _ = _fixture()
    model, clus = tmp_path / "2.vae", tmp_path / "3.clustering"
    model.mkdir(); clus.mkdir()
    fq = os.path.join(GOLDEN, "tenx_clean.fq.gz")
    barcodes = np.array(["AAAA", "AAAC", "AACC", "ACCC"])
    np.savez(model / "latent.npz", X[:4])
    np.savez(model / "barcodes.npz", barcodes)
    args = argparse.Namespace(clusters=2, reads1="", reads2="", interleaved_reads=fq, metaphlan_db="")
    orig = clustering.clustering_rph_kmeans
    clustering.clustering_rph_kmeans = lambda emb, n: np.array([1, 0, 1, 0], dtype=np.int32)
    try:
        clustering.cluster_barcode_reads(args, str(model), str(clus), str(tmp_path))
    finally:
        clustering.clustering_rph_kmeans = orig
    assert (clus / "clusters.tsv").read_text() == "1\tAAAA,AACC\n0\tAAAC,ACCC\n"
    assert np.array_equal(np.load(clus / "clusters.npz")["arr_0"], [1, 0, 1, 0])
    for fn in ("cluster_bin0.fq", "cluster_bin0.barcode", "cluster_bin1.fq", "cluster_bin1.barcode", "clustering_finished"):
        assert (clus / fn).is_file()
    first = (clus / "cluster_bin1.fq").read_text().splitlines()[0]
    assert first.endswith("\tBX:Z:AAAA-1") or first.endswith("\tBX:Z:AACC-1")
