"""``Data`` -- drop-in for /root/reference/src/data.py:8-31 with the normalisation done on the GPU.

Same constructor, ``weights``, ``__len__`` and ``__getitem__ -> {"abd","tnf","bc"}``.  The arithmetic is the
reference's: rows divided by their L1 norm in float64 (all-zero rows stay zero: sklearn ``normalize`` replaces a
zero norm by 1), weights = (row maximum of the normalised abundance)^2 in float64, matrices then narrowed to
float32 (data.py:16-21).  Integer sums and IEEE float64 division make the result bit-identical to the CPU.
The device copies (``abd_dev``/``tnf_dev``) feed ``VAENET.encode`` without a host round trip.
"""
from __future__ import annotations

import logging

import numpy as np
import torch
from torch.utils.data import Dataset


def _l1_rows(x: torch.Tensor) -> torch.Tensor:
    x = x.to(torch.float64)
    norms = x.abs().sum(dim=1, keepdim=True)
    norms = torch.where(norms == 0, torch.ones_like(norms), norms)
    return x / norms


def _as_device_tensor(m, device) -> torch.Tensor:
    if isinstance(m, torch.Tensor):
        return m.to(device)
    m = np.asarray(m)
    if m.dtype == object:
        m = m.astype(np.float64)
    return torch.from_numpy(np.ascontiguousarray(m)).to(device)


def _normalize_counts(m: torch.Tensor, want_weight: bool):
    """(float32 normalised matrix, float64 (row max)^2 or None) of a device int32 count matrix in ONE kernel (pg_normalize_rows)
    instead of a dozen float64 elementwise passes; bit-identical to ``_l1_rows`` + max + cast (the test compares them)"""
    from . import _lib
    from .kmer import _stream_ptr
    m = m.contiguous()
    out = torch.empty(m.shape, dtype=torch.float32, device=m.device)
    w = torch.empty(m.shape[0], dtype=torch.float64, device=m.device) if want_weight else None
    with torch.cuda.device(m.device):
        _lib.check(_lib.load().pg_normalize_rows(m.data_ptr(), m.shape[0], m.shape[1], out.data_ptr(),
                                                 w.data_ptr() if w is not None else None, _stream_ptr(m.device)))
    return out, w


def _is_device_counts(m) -> bool:
    return isinstance(m, torch.Tensor) and m.is_cuda and m.dtype == torch.int32 and m.dim() == 2


class Data(Dataset):
    def __init__(self, barcodes, abd, tnf, device=None):
        super().__init__()
        if device is None:
            device = abd.device if isinstance(abd, torch.Tensor) else ("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(device)
        self.bc = barcodes
        logging.info("calculate sampling weights")
        if _is_device_counts(abd) and _is_device_counts(tnf) and abd.device == self.device and tnf.device == self.device:
            # the count matrices as the feature kernels leave them: one fused pass each
            # (the weights stay on the device until somebody reads ``weights``: the encode of a batch does not, and a copy here
            # would make the host wait for every kernel enqueued so far)
            self.abd_dev, self._weights_dev = _normalize_counts(abd, True)
            self._weights = None if abd.shape[1] else np.zeros(abd.shape[0], dtype=np.float64)
            logging.info("normalize data")
            self.tnf_dev, _ = _normalize_counts(tnf, False)
        else:
            nabd = _l1_rows(_as_device_tensor(abd, self.device))
            if nabd.shape[0]:
                m = nabd.max(dim=1).values
            else:
                m = nabd.new_zeros(0)
            self._weights_dev = m * m
            self._weights = None
            logging.info("normalize data")
            self.abd_dev = nabd.to(torch.float32)
            self.tnf_dev = _l1_rows(_as_device_tensor(tnf, self.device)).to(torch.float32)
        self._abd = self._tnf = None
        logging.info("preprocessing completed")

    # host views, materialised on first use (the reference keeps numpy float32 matrices and the float64 weights here)
    @property
    def weights(self) -> np.ndarray:
        if self._weights is None:
            self._weights = self._weights_dev.cpu().numpy().astype(np.float64)
        return self._weights

    @weights.setter
    def weights(self, value) -> None:
        self._weights = np.asarray(value, dtype=np.float64)

    @property
    def abd(self) -> np.ndarray:
        if self._abd is None:
            self._abd = self.abd_dev.cpu().numpy()
        return self._abd

    @property
    def tnf(self) -> np.ndarray:
        if self._tnf is None:
            self._tnf = self.tnf_dev.cpu().numpy()
        return self._tnf

    def __len__(self):
        return self.abd_dev.shape[0]

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        return {"abd": self.abd[idx, :], "tnf": self.tnf[idx, :], "bc": self.bc[idx]}
