#!/usr/bin/env python3
"""time the two steps that follow the feature path on one GPU: VAE training (epochs over device-resident batches with the
reference's sampling, validation every 100 batches) and RPH-KMeans (n_init = 20) on the latents.
    python tools/bench_train_cluster.py [rows] [epochs] [clusters]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pangaea_amd.clustering import RPHKMeans  # noqa: E402
from pangaea_amd.data import Data  # noqa: E402
from pangaea_amd.loader import shuffled_batches, weighted_batches  # noqa: E402
from pangaea_amd.models.VAENET import VAENET  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
clusters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
np.random.seed(2021)
torch.manual_seed(2021)
# count matrices shaped like the path's output: rows drawn around `clusters` prototype profiles
g = torch.Generator(device=dev).manual_seed(1)
proto_a = torch.rand((clusters, 400), generator=g, device=dev) ** 4
proto_t = torch.rand((clusters, 136), generator=g, device=dev) + 0.2
which = torch.randint(0, clusters, (rows,), generator=g, device=dev)
abd = torch.poisson(proto_a[which] * 300, generator=g).to(torch.int32)
tnf = torch.poisson(proto_t[which] * 400, generator=g).to(torch.int32)
names = np.array([f"bc{i}" for i in range(rows)], dtype=object)
t = time.perf_counter()
data = Data(names, abd, tnf, device=dev)
torch.cuda.synchronize()
print(f"{rows} rows; Data (L1 normalise + weights): {1e3 * (time.perf_counter() - t):.1f} ms")
batch = 2048
train = weighted_batches(data, batch)
test = weighted_batches(data, batch, num_samples=min(int(rows * 0.7), 1_000_000), replacement=False)
orig = shuffled_batches(data, batch)
vae = VAENET(400, 136, 32, clusters, epochs, True, 1, 0.005, 0.2, 0.1, 0.015, 0.0001)
tmp = tempfile.mkdtemp(prefix="pg_train_")
t = time.perf_counter()
vae.train(train, test, orig, tmp, patience=10_000)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"VAE train {epochs} epochs x {len(train)} batches of {batch} (+ validation every 100 batches, encode, files): {dt:.2f} s "
      f"= {1e3 * dt / (epochs * len(train)):.2f} ms per batch all in, {epochs * rows / dt / 1e3:.0f} k rows/s")
latent = np.load(os.path.join(tmp, "latent.npz"))["arr_0"]
for f in os.listdir(tmp):
    os.remove(os.path.join(tmp, f))
os.rmdir(tmp)
t = time.perf_counter()
km = RPHKMeans(n_init=20, n_clusters=clusters)
labels = km.fit_predict(latent)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"RPH-KMeans n_init=20, k={clusters} on {latent.shape}: {dt:.2f} s (skeleton reductions in {km.rp_iter_} rounds, Lloyd {km.n_iter_} iterations in the best run), inertia {km.inertia_:.4g}")
