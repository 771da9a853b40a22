"""Device-resident batches for ``VAENET.train`` -- what ``src/pangaea.py:86-89`` builds with three DataLoaders.

The reference feeds the network through ``DataLoader(dataset, batch_size, num_workers=threads, sampler=...)``: every
item is a Python dict of numpy rows collated on the host.  Here the ``Data`` matrices already live in HBM, so a batch
is one ``index_select`` on the device.  Sampling semantics are kept (utils.py:13-21, pangaea.py:86-89):
  * train:    ``np.random.choice(N, size=N, p=weights/sum, replace=True)`` redrawn every epoch
  * test:     ``np.random.choice(N, size=min(0.7 N, 1e6), p=..., replace=False)`` redrawn every pass
  * original: a fresh ``torch.randperm(N)`` every pass (``shuffle=True``)
so with the same seeds the index sequences are the reference's ("numpy" mode).

"device" mode (the default; ``PANGAEA_SAMPLING=numpy`` or ``mode="numpy"`` selects the other) draws the weighted samples on
the GPU with ``torch.multinomial`` -- with replacement the same distribution, without replacement the same
successive-sampling distribution (numpy draws and renormalises one by one, multinomial ranks exponential keys: both are
sampling without replacement in proportion to the weights) -- from a device generator seeded out of numpy's global
generator, so runs stay reproducible under ``init_all``; only the index SEQUENCE differs from the reference's.  numpy's
draw without replacement over 10^6 rows was half of an epoch's time.
"""
from __future__ import annotations

import os

import numpy as np
import torch


class _Names:
    """the barcodes of a batch, looked up on the host only if somebody asks (training and validation never do)"""

    def __init__(self, bc, sel):
        self._bc, self._sel = bc, sel

    def __array__(self, dtype=None, copy=None):
        sel = self._sel.cpu().numpy() if torch.is_tensor(self._sel) else self._sel
        out = np.asarray(self._bc)[sel]
        return out if dtype is None else out.astype(dtype)

    def __iter__(self):
        return iter(self.__array__())

    def __len__(self):
        return int(self._sel.shape[0])


class DeviceBatches:
    def __init__(self, data, batch_size: int, draw, num_samples: int):
        self.dataset, self.batch_size, self._draw, self.num_samples = data, int(batch_size), draw, int(num_samples)

    def __len__(self) -> int:
        return (self.num_samples + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        drawn = self._draw()
        dev = self.dataset.abd_dev.device
        if torch.is_tensor(drawn):                       # drawn on the device: the indices never visit the host
            idx, idx_host = drawn.to(dev), None
        else:
            idx_host = np.asarray(drawn, dtype=np.int64)
            idx = torch.from_numpy(idx_host).to(dev)
        bc = np.asarray(self.dataset.bc)
        for a in range(0, int(idx.shape[0]), self.batch_size):
            sel = idx[a:a + self.batch_size]
            names = bc[idx_host[a:a + self.batch_size]] if idx_host is not None else _Names(bc, sel)
            yield {"abd": self.dataset.abd_dev.index_select(0, sel), "tnf": self.dataset.tnf_dev.index_select(0, sel), "bc": names}


def sampling_mode(mode: str | None = None) -> str:
    mode = mode or os.environ.get("PANGAEA_SAMPLING", "device")
    if mode not in ("device", "numpy"):
        raise ValueError(f"sampling mode {mode!r} (device or numpy)")
    return mode


def weighted_batches(data, batch_size: int, num_samples: int | None = None, replacement: bool = True, mode: str | None = None) -> DeviceBatches:
    n = len(data)
    num_samples = n if num_samples is None else num_samples
    w = np.asarray(data.weights, dtype=np.float64)
    dev = data.abd_dev.device
    if sampling_mode(mode) == "device" and dev.type == "cuda" and n < (1 << 24):      # (torch.multinomial takes at most 2^24 categories)
        p = torch.from_numpy(w / w.sum()).to(dev)

        def draw_device():
            gen = torch.Generator(device=dev)
            gen.manual_seed(int(np.random.randint(0, 2 ** 31 - 1)))
            return torch.multinomial(p, num_samples, replacement=replacement, generator=gen)

        return DeviceBatches(data, batch_size, draw_device, num_samples)

    def draw():
        # torch.as_tensor(weights, dtype=double) / sum -- WeightedRandomSampler stores float64 weights
        # (an int first argument draws exactly what ``range(0, n)`` draws -- "as if it were np.arange(n)" -- without
        # materialising a million-element Python range first)
        return np.random.choice(n, size=num_samples, p=w / w.sum(), replace=replacement)

    return DeviceBatches(data, batch_size, draw, num_samples)


def shuffled_batches(data, batch_size: int) -> DeviceBatches:
    n = len(data)
    return DeviceBatches(data, batch_size, lambda: torch.randperm(n).numpy(), n)
