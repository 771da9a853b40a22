"""Clustering layer -- counterpart of /root/reference/src/clustering.py:14-129 with RPH-KMeans on the GPU.

``clustering_rph_kmeans(embedding, k)`` keeps the reference's call (``RPHKMeans(n_init=20, n_clusters=k)``,
clustering.py:14-19).  The algorithm is the vendored library's (third_parties/rph_kmeans):
  * bucket width w = half the median of 1 000 random paired distances        (point_reducer_base.py:35-48)
  * repeat until <= 2 000 points: 5 random projections N(0, 1/w) + U(0,1) offsets, truncate to int32, merge
    the points that share all 5 integers into their weighted mean            (point_reducer_cy.py:47-76,
                                                                               _point_reducer_cy_lib.cpp:5-28, .h:52-68)
  * weighted k-means on the skeleton -> initial centres                        (rph_kmeans_.py:116-129)
  * Lloyd iterations on ALL points from those centres, best inertia of n_init (rph_kmeans_.py:143-162)
Where it runs: everything on the device -- projections, bucketing (``torch.unique`` over the 5-integer rows), weighted
merges, the weighted skeleton k-means (greedy k-means++ seeding with sklearn's 2 + ln k local trials, then weighted
Lloyd) and the Lloyd distance / assignment / update steps over all points (the [N,32]x[k,32] distance step is a GEMM +
row argmin).  Nothing but the final labels returns to the host.  Random draws come from numpy's global generator in the
reference's order (``init_all`` seeds it; the device generator of the k-means++ seeding is seeded from it), but bucket numbering follows
``torch.unique`` instead of ``unordered_map`` iteration, so labels are NOT bit-comparable with the reference
 -- parity is by inertia and adjusted Rand index.
"""
from __future__ import annotations

import logging
import math
import os
import warnings
from collections import defaultdict

import numpy as np
import torch

from .utils import run_cmd


def _sq_dists(x: torch.Tensor, c: torch.Tensor, x2: torch.Tensor | None = None) -> torch.Tensor:
    """squared euclidean distances [N, K] in the dtype of x (the distance step)"""
    if x2 is None:
        x2 = (x * x).sum(1, keepdim=True)
    d = x2 - 2.0 * (x @ c.t()) + (c * c).sum(1)[None, :]
    return d.clamp_(min=0)


def _all_reduce(t: torch.Tensor, group, op=None) -> torch.Tensor:
    """in-place all-reduce of a small tensor over the ranks that share the rows (gloo cannot take device tensors: staged)"""
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


@torch.no_grad()
def lloyd(x: torch.Tensor, centers: torch.Tensor, max_iter: int = 300, tol: float = 1e-4, sample_weight: torch.Tensor | None = None,
          group=None, sharded: bool = False):
    """sklearn ``KMeans(init=centers, n_init=1)`` semantics (algorithm "lloyd"): stop when the labels repeat or the
    squared centre shift falls below tol * mean feature variance; an empty cluster takes the point farthest from its
    centre.  Returns (labels int64 [N], centers [K,D], inertia float, n_iter).

    ``sharded`` (rph_kmeans_.py:152-160 runs this on one matrix): ``x`` holds THIS rank's rows only.  Distances and
    assignments are local; per iteration one all-reduce sums the [K, D] coordinate sums and the [K] counts of all ranks
    (K x (D + 1) values: 5 KB for K = 40, D = 32), a second, one-word one agrees on "no label changed anywhere"; the feature
    variance behind the tolerance, the candidates for an empty cluster and the final inertia are reduced the same way.
    Every rank ends with the same centres, inertia and iteration count, and with the labels of its own rows."""
    n, dim = x.shape
    k = centers.shape[0]
    w = torch.ones(n, dtype=x.dtype, device=x.device) if sample_weight is None else sample_weight.to(x)
    x2 = (x * x).sum(1, keepdim=True)
    if sharded:
        import torch.distributed as dist
        mom = torch.cat([x.double().sum(0), (x.double() ** 2).sum(0), torch.tensor([float(n)], dtype=torch.float64, device=x.device)])
        _all_reduce(mom, group)
        n_all = float(mom[-1].item())
        mean = mom[:dim] / n_all
        tol_abs = float((mom[dim:2 * dim] / n_all - mean * mean).clamp(min=0).mean().item()) * tol
    else:
        tol_abs = float(x.var(dim=0, unbiased=False).mean().item()) * tol
    centers = centers.to(x).clone()
    labels_old = None
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        d = _sq_dists(x, centers, x2)
        mind, labels = (d.min(dim=1) if n else (x.new_zeros(0), torch.zeros(0, dtype=torch.int64, device=x.device)))
        sums = torch.zeros((k, dim), dtype=x.dtype, device=x.device).index_add_(0, labels, x * w[:, None])
        cnt = torch.zeros(k, dtype=x.dtype, device=x.device).index_add_(0, labels, w)
        if sharded:
            both = torch.cat([sums, cnt[:, None]], dim=1)
            _all_reduce(both, group)
            sums, cnt = both[:, :dim].contiguous(), both[:, dim].contiguous()
        empty = torch.nonzero(cnt == 0).flatten()
        if empty.numel() and not sharded:                   # relocate empty clusters to the farthest points
            far = torch.argsort(mind, descending=True)[:empty.numel()]
            for e, f in zip(empty.tolist(), far.tolist()):
                old = int(labels[f])
                sums[old] -= x[f] * w[f]; cnt[old] -= w[f]
                sums[e] = x[f] * w[f]; cnt[e] = w[f]
                labels[f] = e
        elif empty.numel():
            # the same, over all ranks' rows: every rank offers its own farthest points (distance, weight, label, owner, index,
            # coordinates), the offers are gathered and every rank applies the same relocations to the reduced sums; the owner
            # of a relocated point also changes that point's label
            n_e = int(empty.numel())
            me, world = dist.get_rank(group), dist.get_world_size(group)
            offer = torch.full((n_e, 5 + dim), -1.0, dtype=torch.float64, device=x.device)
            take = min(n_e, n)
            if take:
                far = torch.argsort(mind, descending=True)[:take]
                offer[:take, 0] = mind[far].double(); offer[:take, 1] = w[far].double(); offer[:take, 2] = labels[far].double()
                offer[:take, 3] = float(me); offer[:take, 4] = far.double(); offer[:take, 5:] = x[far].double()
            offers = torch.zeros((world, n_e, 5 + dim), dtype=torch.float64, device=x.device)
            offers[me] = offer
            _all_reduce(offers, group)                      # (a sum over one-hot rank slots = an all-gather, staged like the rest)
            flat = offers.view(-1, 5 + dim)
            order = torch.argsort(flat[:, 0], descending=True, stable=True)[:n_e]
            for e, row in zip(empty.tolist(), flat[order].tolist()):
                if row[0] < 0:
                    continue                                # (fewer points than empty clusters)
                old, wf = int(row[2]), row[1]
                xf = torch.tensor(row[5:], dtype=x.dtype, device=x.device)
                sums[old] -= xf * wf; cnt[old] -= wf
                sums[e] = xf * wf; cnt[e] = wf
                if int(row[3]) == me:
                    labels[int(row[4])] = e
        new_centers = sums / cnt.clamp(min=1e-30)[:, None]
        shift = float(((new_centers - centers) ** 2).sum().item())
        centers = new_centers
        same = labels_old is not None and torch.equal(labels, labels_old)
        if sharded:
            flag = torch.tensor([1 if same else 0], dtype=torch.int32, device=x.device)
            same = bool(_all_reduce(flag, group, dist.ReduceOp.MIN).item())
        if same:
            break
        labels_old = labels
        if shift <= tol_abs:
            break
    d = _sq_dists(x, centers, x2)
    mind, labels = (d.min(dim=1) if n else (x.new_zeros(0), torch.zeros(0, dtype=torch.int64, device=x.device)))
    total = (mind * w).sum().double().reshape(1)
    if sharded:
        _all_reduce(total, group)
    return labels, centers, float(total.item()), n_iter


@torch.no_grad()
def kmeans_plusplus(x: torch.Tensor, w: torch.Tensor, k: int, gen: torch.Generator) -> torch.Tensor:
    """greedy k-means++ seeding with sample weights, as sklearn's ``_kmeans_plusplus`` (the seeding inside the reference's
    ``KMeans(n_clusters=k).fit_predict(reduced_X, sample_weight)``, rph_kmeans_.py:121-124): the first centre is drawn in
    proportion to the weights, every further one is the best of 2 + ln k candidates drawn in proportion to weight x squared
    distance to the nearest centre so far.  Returns centres [k, D]."""
    n_trials = 2 + int(math.log(k))
    centers = torch.empty((k, x.shape[1]), dtype=x.dtype, device=x.device)
    first = torch.multinomial(w / w.sum(), 1, generator=gen)
    centers[0] = x[first[0]]
    x2 = (x * x).sum(1, keepdim=True)
    closest = _sq_dists(x, centers[0:1], x2).squeeze(1)
    for c in range(1, k):
        p = closest * w
        total = p.sum()
        # (every point already a centre: fall back to the weights, as sklearn's searchsorted on a flat cumsum would)
        p = torch.where(total > 0, p / total.clamp(min=1e-300), w / w.sum())
        cand = torch.multinomial(p, n_trials, replacement=True, generator=gen)
        d = torch.minimum(_sq_dists(x, x[cand], x2), closest[:, None])          # [G, n_trials]
        best = (d * w[:, None]).sum(0).argmin()
        closest = d[:, best]
        centers[c] = x[cand[best]]
    return centers


@torch.no_grad()
def weighted_kmeans(x: torch.Tensor, w: torch.Tensor, k: int, n_init: int = 1, max_iter: int = 300, tol: float = 1e-4,
                    gen: torch.Generator | None = None):
    """weighted k-means on the device (the skeleton step): best of ``n_init`` k-means++ seedings + Lloyd runs.
    Returns (centers [k,D], labels int64 [G], inertia)."""
    if gen is None:
        gen = torch.Generator(device=x.device)
        gen.manual_seed(int(np.random.randint(0, 2 ** 31 - 1)))
    best = None
    for _ in range(max(1, n_init)):
        c0 = kmeans_plusplus(x, w, k, gen)
        labels, centers, inertia, _ = lloyd(x, c0, max_iter=max_iter, tol=tol, sample_weight=w)
        if best is None or inertia < best[2]:
            best = (centers, labels, inertia)
    return best


class RPHKMeans:
    """device-resident ``rph_kmeans.RPHKMeans`` (constructor arguments and fitted attributes as the library's)"""

    def __init__(self, n_clusters=8, n_init=1, w=None, max_point=2000, proj_num=5, max_iter=1000, sample_dist_num=1000,
                 verbose=0, device=None, skeleton_n_init=1):
        self.n_clusters, self.n_init, self.w = n_clusters, n_init, w
        self.skeleton_n_init = skeleton_n_init          # k-means++ restarts of the skeleton step (sklearn's n_init="auto" is 1)
        self.max_point, self.proj_num, self.max_iter, self.sample_dist_num = max_point, proj_num, max_iter, int(sample_dist_num)
        self.verbose = verbose
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("RPHKMeans runs on the GPU; pass device='cpu' explicitly to run the same torch ops on the host")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.inertia_ = self.cluster_centers_ = self.labels_ = self.n_iter_ = None
        self.reduced_X_ = self.reduced_X_weight_ = self.rp_labels_ = self.rp_iter_ = self.init_centers_ = None

    # -------------------------------------------------------------- point reduction

    def _get_w(self, x: torch.Tensor) -> float:
        if self.w is not None:
            return self.w
        n = x.shape[0]
        a = torch.from_numpy(np.random.choice(n, self.sample_dist_num)).to(x.device)
        b = torch.from_numpy(np.random.choice(n, self.sample_dist_num)).to(x.device)
        dist = (x[a].double() - x[b].double()).pow(2).sum(1).sqrt()
        self.w = float(np.median(dist.cpu().numpy()) * 0.5)
        return self.w

    @torch.no_grad()
    def reduce_points(self, x: torch.Tensor):
        """(reduced_X [G,D], weight [G], labels int64 [N] -> row of reduced_X, iterations)"""
        n, dim = x.shape
        w = self._get_w(x)
        if not (w > 0):
            raise RuntimeError("RPH bucket width is zero: the sampled points are all identical")
        labels = torch.arange(n, device=x.device)
        weight = torch.ones(n, dtype=x.dtype, device=x.device)
        red = x
        it = 0
        while it < self.max_iter and red.shape[0] > self.max_point:
            b = np.random.uniform(0, 1, size=(self.proj_num,))
            proj = np.random.normal(0, 1.0 / w, size=(dim, self.proj_num))
            pj = (red.double() @ torch.from_numpy(proj).to(x.device) + torch.from_numpy(b).to(x.device)).to(torch.int32)   # trunc toward 0
            _, bucket = torch.unique(pj, dim=0, return_inverse=True)
            g = int(bucket.max().item()) + 1
            new_w = torch.zeros(g, dtype=x.dtype, device=x.device).index_add_(0, bucket, weight)
            new_x = torch.zeros((g, dim), dtype=x.dtype, device=x.device).index_add_(0, bucket, red * weight[:, None])
            red = new_x / new_w[:, None]
            weight = new_w
            labels = bucket[labels]
            it += 1
        return red, weight, labels, it

    def init_centers(self, x: torch.Tensor):
        """(initial centres, reduced points, their weights, sample -> reduced point, reducer iterations, skeleton inertia,
        skeleton labels) -- the weighted skeleton k-means of rph_kmeans_.py:116-129, on the device"""
        red, weight, labels, it = self.reduce_points(x)
        if red.shape[0] < self.n_clusters:
            raise RuntimeError("Number of reduced points is too small, please try smaller w or larger proj_num")
        centers, pred, inertia = weighted_kmeans(red, weight, self.n_clusters, n_init=self.skeleton_n_init)
        return centers, red, weight, labels, it, inertia, pred

    # -------------------------------------------------------------- fit

    def fit(self, X):
        x = torch.as_tensor(np.ascontiguousarray(X) if isinstance(X, np.ndarray) else X).to(self.device)
        if x.dtype not in (torch.float32, torch.float64):
            x = x.float()
        self.inertia_ = np.inf
        for _ in range(self.n_init):
            centers0, red, weight, rp_labels, rp_iter, _, _ = self.init_centers(x)
            labels, centers, inertia, n_iter = lloyd(x, centers0.to(x))
            if inertia < self.inertia_:
                self.inertia_ = inertia
                self.labels_, self.cluster_centers_, self.n_iter_ = labels.cpu().numpy().astype(np.int32), centers.cpu().numpy(), n_iter
                self.init_centers_, self.reduced_X_, self.reduced_X_weight_ = centers0.cpu().numpy(), red.cpu().numpy(), weight.cpu().numpy()
                self.rp_labels_, self.rp_iter_ = rp_labels.cpu().numpy(), rp_iter
        return self

    def fit_predict(self, X):
        return self.fit(X).labels_

    def predict(self, X):
        x = torch.as_tensor(X).to(self.device)
        c = torch.from_numpy(self.cluster_centers_).to(x)
        return _sq_dists(x, c).argmin(1).cpu().numpy().astype(np.int32)


def clustering_rph_kmeans(embedding, k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clt = RPHKMeans(n_init=20, n_clusters=k, verbose=0)
        return clt.fit_predict(embedding)


@torch.no_grad()
def clustering_rph_kmeans_sharded(mu_local: torch.Tensor, mu_all, k: int, n_init: int = 20, group=None, device=None):
    """``clustering_rph_kmeans`` with the latent rows sharded over the ranks of ``group``: rank 0, which holds every
    row anyway (it writes latent.npz), reduces the points and seeds the centres exactly as the one-process fit does (same
    numpy draws in the same order); the centres are broadcast (K x 32 floats) and the Lloyd iterations over ALL points --
    rph_kmeans_.py:152-160 -- run on every rank's own rows with the per-iteration all-reduce of ``lloyd(sharded=True)``.
    Returns the labels of all rows in rank order on rank 0 (int32, as the one-process call), None elsewhere."""
    import torch.distributed as dist
    from . import dist as pdist
    me = dist.get_rank(group)
    x = torch.as_tensor(mu_local)
    if device is not None:
        x = x.to(device)
    if x.dtype not in (torch.float32, torch.float64):
        x = x.float()
    clt = None
    if me == 0:
        clt = RPHKMeans(n_init=n_init, n_clusters=k, verbose=0, device=x.device)
        full = torch.as_tensor(mu_all).to(x)
    best = None
    for _ in range(n_init):
        centers0 = clt.init_centers(full)[0].to(x).contiguous() if me == 0 else torch.empty((k, x.shape[1]), dtype=x.dtype, device=x.device)
        if x.is_cuda and dist.get_backend(group) == "gloo":
            h = centers0.cpu()
            dist.broadcast(h, src=0, group=group)
            centers0.copy_(h)
        else:
            dist.broadcast(centers0, src=0, group=group)
        labels, _, inertia, _ = lloyd(x, centers0, group=group, sharded=True)
        if best is None or inertia < best[1]:               # (the inertia is the reduced one: the same choice on every rank)
            best = (labels, inertia)
    got = pdist.gather_rows(best[0].to(torch.int32)[:, None], dst=0, group=group)
    return got[:, 0].cpu().numpy().astype(np.int32) if me == 0 else None


def write_clusters_tsv(path: str, clusters, barcodes) -> None:
    """``<label>\\t<bc1>,<bc2>,...`` per label in first-seen order (clustering.py:107-112)"""
    cluster2barcodes = defaultdict(list)
    for lab, bc in zip(clusters, barcodes):
        cluster2barcodes[lab].append(str(bc))
    with open(path, "w") as tsv:
        for cluster_id in cluster2barcodes:
            tsv.write("{}\t{}\n".format(cluster_id, ",".join(cluster2barcodes[cluster_id])))


def cluster_barcode_reads(args, model_path, cluster_path, script_path, clusters=None):
    """step 3 of pangaea.py: latent.npz/barcodes.npz -> clusters.npz, clusters.tsv, cluster_bin<label>.{fq,barcode},
    clustering_finished -- the on-disk bin layout the unchanged reassembly stage reads (bin_assembly.sh:18).
    ``clusters``: labels already computed for the rows of latent.npz (the sharded Lloyd of a multi-rank run)."""
    from .binwriter import extract_reads
    output_npz = os.path.join(cluster_path, "clusters.npz")
    output_tsv = os.path.join(cluster_path, "clusters.tsv")
    embedding_path = os.path.join(model_path, "latent.npz")
    barcodes_path = os.path.join(model_path, "barcodes.npz")
    if not os.path.isfile(embedding_path) or not os.path.isfile(barcodes_path):
        raise FileNotFoundError(f"{embedding_path} or {barcodes_path} not found")
    if not os.path.isfile(output_tsv) and clusters is not None:
        barcodes = np.load(barcodes_path)["arr_0"]
        assert len(clusters) == len(barcodes)
        np.savez(output_npz, clusters)
        logging.info("saving clustering tsv")
        write_clusters_tsv(output_tsv, clusters, barcodes)
    elif not os.path.isfile(output_tsv):
        embedding = np.load(embedding_path)["arr_0"]
        barcodes = np.load(barcodes_path)["arr_0"]
        if args.clusters:
            num_classes = args.clusters
        else:
            # the reference estimates 8 x Shannon diversity with metaphlan (clustering.py:92-102): an external tool,
            # outside this path -- run the reference's own script when it is installed next to us
            diversity = os.path.join(script_path, "scripts", "calculate_diversity.sh")
            if not os.path.isfile(diversity):
                raise FileNotFoundError("number of clusters (-c) not given and scripts/calculate_diversity.sh not found")
            reads = args.reads1 if (args.reads1 and args.reads2) else args.interleaved_reads
            run_cmd([diversity, reads, args.metaphlan_db, cluster_path])
            shannon = np.loadtxt(os.path.join(cluster_path, "metaphlan_tmp/diversity_analysis/profiles_table_shannon.txt"))
            num_classes = int(8 * shannon)
            logging.info(f"estimated num_classes: {num_classes}")
            run_cmd(["rm", "-rf", os.path.join(cluster_path, "metaphlan_tmp")])
        clusters = clustering_rph_kmeans(embedding, num_classes)
        np.savez(output_npz, clusters)
        logging.info("saving clustering tsv")
        write_clusters_tsv(output_tsv, clusters, barcodes)
    else:
        logging.info("existing clustering result found")
    prefix = os.path.join(cluster_path, "cluster")
    if args.reads1 and args.reads2:
        extract_reads(args.reads1, args.reads2, output_tsv, prefix)
    elif args.interleaved_reads:
        extract_reads(args.interleaved_reads, None, output_tsv, prefix)
    else:
        logging.error("no reads provided")
        raise FileNotFoundError("no reads provided")
    with open(os.path.join(cluster_path, "clustering_finished"), "w") as f:
        f.write("finished")
