"""Device-resident batches for ``VAENET.train`` -- what ``src/pangaea.py:86-89`` builds with three DataLoaders.

The reference feeds the network through ``DataLoader(dataset, batch_size, num_workers=threads, sampler=...)``: every
item is a Python dict of numpy rows collated on the host.  Here the ``Data`` matrices already live in HBM, so a batch
is one ``index_select`` on the device.  Sampling semantics are kept (utils.py:13-21, pangaea.py:86-89):
  * train:    ``np.random.choice(N, size=N, p=weights/sum, replace=True)`` redrawn every epoch
  * test:     ``np.random.choice(N, size=min(0.7 N, 1e6), p=..., replace=False)`` redrawn every pass
  * original: a fresh ``torch.randperm(N)`` every pass (``shuffle=True``)
so with the same seeds the index sequences are the reference's.
"""
from __future__ import annotations

import numpy as np
import torch


class DeviceBatches:
    def __init__(self, data, batch_size: int, draw, num_samples: int):
        self.dataset, self.batch_size, self._draw, self.num_samples = data, int(batch_size), draw, int(num_samples)

    def __len__(self) -> int:
        return (self.num_samples + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        idx_host = np.asarray(self._draw(), dtype=np.int64)
        idx = torch.from_numpy(idx_host).to(self.dataset.abd_dev.device)
        bc = np.asarray(self.dataset.bc)
        for a in range(0, len(idx_host), self.batch_size):
            sel = idx[a:a + self.batch_size]
            yield {"abd": self.dataset.abd_dev.index_select(0, sel), "tnf": self.dataset.tnf_dev.index_select(0, sel),
                   "bc": bc[idx_host[a:a + self.batch_size]]}


def weighted_batches(data, batch_size: int, num_samples: int | None = None, replacement: bool = True) -> DeviceBatches:
    n = len(data)
    num_samples = n if num_samples is None else num_samples
    w = np.asarray(data.weights, dtype=np.float64)

    def draw():
        # torch.as_tensor(weights, dtype=double) / sum -- WeightedRandomSampler stores float64 weights
        # (an int first argument draws exactly what ``range(0, n)`` draws -- "as if it were np.arange(n)" -- without
        # materialising a million-element Python range first)
        return np.random.choice(n, size=num_samples, p=w / w.sum(), replace=replacement)

    return DeviceBatches(data, batch_size, draw, num_samples)


def shuffled_batches(data, batch_size: int) -> DeviceBatches:
    n = len(data)
    return DeviceBatches(data, batch_size, lambda: torch.randperm(n).numpy(), n)
