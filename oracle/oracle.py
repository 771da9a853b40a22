"""ctypes face of liboracle.so + the numpy / torch-CPU restatements of the Python half of the path.

TEST INFRASTRUCTURE ONLY -- importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never from ``pangaea_amd``.  See the header of
``pangaea_oracle.c`` for what is restated and how it is pinned to the reference.

Python-side restatements (reference file:line under /root/reference/src):
  * :func:`data_normalize`   -- ``data.py:16-21``   (sklearn ``normalize(.,"l1")`` in fp64 -> fp32, weights)
  * :func:`vae_embedding`    -- ``models/VAENET.py:232-236`` with the layer stack of ``:201-210``
  * :func:`vae_forward_loss` -- ``models/VAENET.py:222-253`` + ``:161-184`` (training forward, given epsilon)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(ref: bool = True) -> None:
    """compile liboracle.so (and oracle/_ref when the reference sources are present)"""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("PG_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # PG_ORACLE_LIB: sanitizer builds
    if not os.path.exists(path):
        build(ref=False)
    L = C.CDLL(path)
    vp, i64, cp = C.c_void_p, C.c_int64, C.c_char_p
    sig = {
        "orc_parse_fastq": (vp, [cp, cp]),
        "orc_reads_free": (None, [vp]),
        "orc_reads_n_runs": (i64, [vp]),
        "orc_reads_n_pairs": (i64, [vp]),
        "orc_reads_n_unpaired": (i64, [vp]),
        "orc_reads_mode": (C.c_int, [vp]),
        "orc_reads_seq": (vp, [vp]),
        "orc_reads_seq_off": (C.POINTER(i64), [vp]),
        "orc_reads_name": (cp, [vp, i64]),
        "orc_reads_all_seq": (vp, [vp]),
        "orc_reads_all_len": (i64, [vp]),
        "orc_run_survives": (C.c_int, [vp, i64, C.c_int]),
        "orc_revcomp": (C.c_uint64, [C.c_uint64, C.c_int]),
        "orc_tnf_ncols": (C.c_int, [C.c_int]),
        "orc_tnf_columns": (C.c_int, [C.c_int, vp]),
        "orc_tnf_row": (C.c_int, [cp, i64, C.c_int, vp]),
        "orc_table_new": (vp, [C.c_int, C.c_int]),
        "orc_table_free": (None, [vp]),
        "orc_table_count_seq": (C.c_int, [vp, vp, i64, C.c_int]),
        "orc_table_count_known": (C.c_int, [vp, vp, i64, C.c_int]),
        "orc_table_size": (i64, [vp]),
        "orc_table_get": (C.c_uint64, [vp, C.c_uint64, C.POINTER(C.c_int)]),
        "orc_table_export": (None, [vp, vp, vp]),
        "orc_table_set": (C.c_int, [vp, C.c_uint64, C.c_uint64]),
        "orc_table_dump": (C.c_int, [vp, cp]),
        "orc_table_load_dump": (vp, [cp, C.c_int, C.c_int]),
        "orc_abd_row": (C.c_int, [cp, i64, C.c_int, vp, C.c_int, C.c_int, vp]),
        "orc_features": (i64, [vp, C.c_int, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_int]),
        "orc_write_csv_gz": (C.c_int, [cp, cp, vp, i64, i64]),
        "orc_num_threads": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


def ref_tool(name: str) -> str | None:
    """path of a prebuilt reference binary under oracle/_ref, or None"""
    p = os.path.join(_HERE, "_ref", name)
    return p if os.access(p, os.X_OK) else None


def _b(s) -> bytes:
    return s if isinstance(s, bytes) else str(s).encode()


# ----------------------------------------------------------------------------- runs


class Reads:
    """every run of a barcode-sorted FASTQ, assembled the way the reference counters do it"""

    def __init__(self, r1: str, r2: str | None = None):
        self._L = lib()
        self._h = self._L.orc_parse_fastq(_b(r1), _b(r2) if r2 else None)
        if not self._h:
            raise RuntimeError(f"oracle: cannot parse {r1!r} (the reference would abort on this input)")
        n = self.n_runs = self._L.orc_reads_n_runs(self._h)
        self.n_pairs = self._L.orc_reads_n_pairs(self._h)
        self.n_unpaired = self._L.orc_reads_n_unpaired(self._h)
        self.mode = {0: "", 1: "10x", 2: "stLFR"}[self._L.orc_reads_mode(self._h)]
        off = self._L.orc_reads_seq_off(self._h)
        self.seq_off = np.array([off[i] for i in range(n + 1)], dtype=np.int64)
        self.names = [self._L.orc_reads_name(self._h, i).decode() for i in range(n)]

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_reads_free(self._h)
            self._h = None

    def seq(self, i: int) -> bytes:
        base = self._L.orc_reads_seq(self._h)
        a, b = int(self.seq_off[i]), int(self.seq_off[i + 1])
        return C.string_at(base + a, b - a)

    def all_seq(self) -> bytes:
        return C.string_at(self._L.orc_reads_all_seq(self._h), self._L.orc_reads_all_len(self._h))

    def surviving(self, min_len: int) -> list[int]:
        return [i for i in range(self.n_runs) if self._L.orc_run_survives(self._h, i, min_len)]

    def features(self, min_len: int, k_tnf: int | None = 4, k_abd: int | None = None, table: "Table | None" = None,
                 window: int = 10, vsize: int = 400, threads: int = 1):
        """(names[N], tnf int64 [N, D] or None, abd int64 [N, V] or None) for the surviving runs"""
        L = self._L
        n = L.orc_features(self._h, min_len, 0, None, 0, None, 1, 1, None, None, 1)
        rows = np.zeros(n, dtype=np.int64)
        tnf = np.zeros((n, tnf_ncols(k_tnf)), dtype=np.int64) if k_tnf else None
        abd = np.zeros((n, vsize), dtype=np.int64) if k_abd else None
        got = L.orc_features(
            self._h, min_len,
            k_tnf or 0, tnf.ctypes.data if tnf is not None else None,
            k_abd or 0, table._h if table is not None else None, window, vsize,
            abd.ctypes.data if abd is not None else None,
            rows.ctypes.data, threads)
        if got != n:
            raise RuntimeError("oracle: orc_features failed")
        return [self.names[i] for i in rows], tnf, abd


# ----------------------------------------------------------------------------- codes


def revcomp(code: int, k: int) -> int:
    return int(lib().orc_revcomp(code, k))


def tnf_ncols(k: int) -> int:
    n = lib().orc_tnf_ncols(k)
    if n < 0:
        raise ValueError(f"oracle: tnf k={k} unsupported")
    return n


def tnf_columns(k: int) -> np.ndarray:
    out = np.zeros(tnf_ncols(k), dtype=np.uint32)
    lib().orc_tnf_columns(k, out.ctypes.data)
    return out


def code_to_kmer(code: int, k: int) -> str:
    return "".join("ACTG"[(code >> (2 * (k - 1 - j))) & 3] for j in range(k))


def tnf_row(seq: bytes, k: int) -> np.ndarray:
    out = np.zeros(tnf_ncols(k), dtype=np.int64)
    lib().orc_tnf_row(seq, len(seq), k, out.ctypes.data)
    return out


# ----------------------------------------------------------------------------- table


class Table:
    """exact canonical k-mer multiplicities (jellyfish count -C stand-in) or a loaded dump"""

    def __init__(self, k: int, threads: int = 1, _h=None):
        self._L = lib()
        self.k = k
        self._h = _h if _h is not None else self._L.orc_table_new(k, threads)
        if not self._h:
            raise ValueError(f"oracle: cannot make a table for k={k}")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_table_free(self._h)
            self._h = None

    @classmethod
    def from_dump(cls, path: str, k: int, threads: int = 1) -> "Table":
        h = lib().orc_table_load_dump(_b(path), k, threads)
        if not h:
            raise RuntimeError(f"oracle: cannot load dump {path}")
        return cls(k, _h=h)

    def count(self, seq: bytes | np.ndarray, lowercase_is_base: bool = False) -> "Table":
        if isinstance(seq, np.ndarray):
            ptr, n = seq.ctypes.data, seq.size
        else:
            buf = C.create_string_buffer(seq, len(seq))
            ptr, n = C.addressof(buf), len(seq)
        if self._L.orc_table_count_seq(self._h, ptr, n, int(lowercase_is_base)):
            raise MemoryError("oracle: table growth failed")
        return self

    def count_known(self, seq: bytes | np.ndarray, lowercase_is_base: bool = False) -> "Table":
        """add the occurrences in ``seq`` of the k-mers that are keys already; no key is created.  ``zeroed_keys_of`` + this =
        exact global multiplicities of a chosen few k-mers over a text too large to count whole"""
        if isinstance(seq, np.ndarray):
            ptr, n = seq.ctypes.data, seq.size
        else:
            buf = C.create_string_buffer(seq, len(seq))
            ptr, n = C.addressof(buf), len(seq)
        self._L.orc_table_count_known(self._h, ptr, n, int(lowercase_is_base))
        return self

    @classmethod
    def zeroed_keys_of(cls, k: int, seqs, threads: int = 1) -> "Table":
        """table whose keys are the canonical k-mers of the given texts, all with count 0"""
        t = cls(k, threads)
        for q in seqs:
            t.count(q)
        for key in t.items()[0]:
            t.set(int(key), 0)
        return t

    def __len__(self) -> int:
        return int(self._L.orc_table_size(self._h))

    def get(self, canon: int) -> int | None:
        found = C.c_int(0)
        v = self._L.orc_table_get(self._h, canon, C.byref(found))
        return int(v) if found.value else None

    def set(self, canon: int, count: int) -> None:
        self._L.orc_table_set(self._h, canon, count)

    def items(self):
        n = len(self)
        keys = np.zeros(n, dtype=np.uint64)
        vals = np.zeros(n, dtype=np.uint64)
        self._L.orc_table_export(self._h, keys.ctypes.data, vals.ctypes.data)
        order = np.argsort(keys)
        return keys[order], vals[order]

    def dump(self, path: str) -> None:
        if self._L.orc_table_dump(self._h, _b(path)):
            raise OSError(f"oracle: cannot write {path}")


def abd_row(seq: bytes, k: int, table: Table, window: int = 10, vsize: int = 400) -> np.ndarray:
    out = np.zeros(vsize, dtype=np.int64)
    if lib().orc_abd_row(seq, len(seq), k, table._h, window, vsize, out.ctypes.data):
        raise ValueError("oracle: bad abundance parameters")
    return out


def write_csv_gz(path: str, names, mat: np.ndarray) -> None:
    mat = np.ascontiguousarray(mat, dtype=np.int64)
    blob = b"".join(_b(n) + b"\0" for n in names)
    if lib().orc_write_csv_gz(_b(path), blob, mat.ctypes.data, mat.shape[0], mat.shape[1]):
        raise OSError(f"oracle: cannot write {path}")


def num_threads() -> int:
    return int(lib().orc_num_threads())


# ----------------------------------------------------------------------------- data.py / VAENET.py halves


def data_normalize(abd: np.ndarray, tnf: np.ndarray):
    """``Data.__init__``: rows / L1 norm in float64 (all-zero rows stay zero), weights = (row max)^2 in
    float64, then the matrices narrowed to float32."""
    def l1(x):
        x = np.asarray(x, dtype=np.float64)
        norms = np.abs(x).sum(axis=1)
        norms[norms == 0.0] = 1.0
        return x / norms[:, None]

    nabd = l1(abd)
    weights = nabd.max(axis=1) ** 2 if nabd.shape[0] else np.zeros(0)
    return nabd.astype(np.float32), l1(tnf).astype(np.float32), weights.astype(np.float64)


def vae_embedding(state: dict, abd: np.ndarray, tnf: np.ndarray) -> np.ndarray:
    """eval-mode ``emebdding``: cat -> [Linear, BatchNorm1d(running stats, eps 1e-5), LeakyReLU(slope 1.0 =
    identity), Dropout(eval = identity)] x 2 -> l_mu.  Plain torch fp32 on the CPU."""
    import torch
    import torch.nn.functional as F

    t = {k: torch.as_tensor(np.asarray(v)) for k, v in state.items()}
    x = torch.cat([torch.as_tensor(abd), torch.as_tensor(tnf)], dim=1).float()
    for lin, bn in (("encoder.0", "encoder.1"), ("encoder.4", "encoder.5")):
        x = F.linear(x, t[lin + ".weight"], t[lin + ".bias"])
        x = F.batch_norm(x, t[bn + ".running_mean"], t[bn + ".running_var"], t[bn + ".weight"], t[bn + ".bias"],
                         training=False, eps=1e-5)
        x = F.leaky_relu(x, negative_slope=1.0)
    return F.linear(x, t["l_mu.weight"], t["l_mu.bias"]).numpy()


def vae_forward_loss(state: dict, abd: np.ndarray, tnf: np.ndarray, epsilon: np.ndarray, wa: float, wt: float, w_kl: float):
    """eval-mode training forward (``forward`` + ``unlabeled_loss``): encoder -> mu, logsigma = softplus(l_sigma(h)),
    z = mu + eps * exp(logsigma / 2), decoder (same layer pattern) -> output -> softmax over the two blocks; losses =
    cross entropies with eps 1e-9, KL = -1/2 sum(1 + logsigma - mu^2 - exp(logsigma)), total = wa*abd + wt*tnf + w_kl*KL."""
    import torch
    import torch.nn.functional as F

    t = {k: torch.as_tensor(np.asarray(v)) for k, v in state.items()}

    def stack(x, prefix):
        for lin, bn in ((prefix + ".0", prefix + ".1"), (prefix + ".4", prefix + ".5")):
            x = F.linear(x, t[lin + ".weight"], t[lin + ".bias"])
            x = F.batch_norm(x, t[bn + ".running_mean"], t[bn + ".running_var"], t[bn + ".weight"], t[bn + ".bias"], training=False, eps=1e-5)
        return x

    a, b = torch.as_tensor(abd).float(), torch.as_tensor(tnf).float()
    h = stack(torch.cat([a, b], 1), "encoder")
    mu = F.linear(h, t["l_mu.weight"], t["l_mu.bias"])
    logsigma = F.softplus(F.linear(h, t["l_sigma.weight"], t["l_sigma.bias"]))
    z = mu + torch.as_tensor(epsilon) * torch.exp(logsigma / 2)
    out = F.linear(stack(z, "decoder"), t["output.weight"], t["output.bias"])
    n_abd = a.shape[1]
    abd_rec, tnf_rec = F.softmax(out[:, :n_abd], 1), F.softmax(out[:, n_abd:], 1)
    l_abd = -(torch.log(abd_rec + 1e-9) * a).sum(-1).mean()
    l_tnf = -(torch.log(tnf_rec + 1e-9) * b).sum(-1).mean()
    l_kl = (-0.5 * (1 + logsigma - mu.pow(2) - logsigma.exp()).sum(1)).mean()
    return {"mu": mu.numpy(), "logsigma": logsigma.numpy(), "abd_rec": abd_rec.numpy(), "tnf_rec": tnf_rec.numpy(),
            "abd": float(l_abd), "tnf": float(l_tnf), "kl": float(l_kl), "total": float(wa * l_abd + wt * l_tnf + w_kl * l_kl)}
