"""Seeded synthetic barcoded read pairs, generated directly in packed stream form on any torch device.

Bench/test support: a community of ``n_genomes`` random genomes with log-normal abundances; every
barcode draws one genome and a ``fragment`` bp window, its ``pairs_per_barcode`` pairs fall uniformly inside the
window with inserts ~U(300,500), mate 2 reverse-complemented; ``sub_rate`` substitutions per base, ``n_rate`` of
the reads carry one N, the last ``unbarcoded`` fraction of the pairs has no barcode (the sorted tail of
run_pangaea:248).  All randomness is integer hashing on int64 tensors, so CPU and GPU produce identical streams.
The genome itself is never stored: base(g, i) = hash(seed, g, i) & 3.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from .reads import ReadStream, words_for

_M1 = -0x61c8864680b583eb        # 0x9e3779b97f4a7c15 as int64
_M2 = -0x40a7b892e31b1a47        # 0xbf58476d1ce4e5b9
_M3 = -0x6b2fb644ecceee15        # 0x94d049bb133111eb


def _lsr(x: torch.Tensor, s: int) -> torch.Tensor:
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 (two's-complement wrap-around is identical on CPU and GPU)"""
    x = x + _M1
    x = (x ^ _lsr(x, 30)) * _M2
    x = (x ^ _lsr(x, 27)) * _M3
    return x ^ _lsr(x, 31)


def _unit(h: torch.Tensor) -> torch.Tensor:
    """hash -> float64 in [0,1)"""
    return _lsr(h, 11).to(torch.float64) * (1.0 / (1 << 53))


@dataclass
class SynthConfig:
    n_pairs: int
    n_barcodes: int
    read_len: int = 150
    n_genomes: int = 64
    genome_len: int = 2_000_000
    fragment: int = 50_000
    sub_rate: float = 0.001
    n_rate: float = 0.005
    unbarcoded: float = 0.02
    seed: int = 2021
    first_pair: int = 0          # global index of this shard's first pair (multi-GPU shards of one data set)
    poisson_mean: float = 0.0    # > 0: pairs per barcode ~ Poisson(mean) (at least 1) instead of a fixed number -- the long reads of
                                 # the hybrid mode, whose names serve as barcodes (assign_barcodes.cpp:156); n_barcodes then follows

    @property
    def chars_per_pair(self) -> int:
        return 2 * (self.read_len + 1)

    @property
    def n_barcoded_pairs(self) -> int:
        return self.n_pairs - int(self.n_pairs * self.unbarcoded)

    @property
    def pairs_per_barcode(self) -> int:
        return max(1, self.n_barcoded_pairs // self.n_barcodes)


_BOUNDS_CACHE: dict = {}


def barcode_bounds(cfg: SynthConfig) -> np.ndarray | None:
    """Poisson mode: cumulative pair counts [n_barcodes + 1] of the barcoded pairs (None in the fixed-size mode)"""
    if cfg.poisson_mean <= 0:
        return None
    if cfg.first_pair:
        raise ValueError("Poisson barcode sizes are for single-shard data sets")
    key = (cfg.seed, cfg.poisson_mean, cfg.n_barcoded_pairs)
    if key not in _BOUNDS_CACHE:
        rs = np.random.RandomState(cfg.seed + 7)
        target = cfg.n_barcoded_pairs
        sizes = np.maximum(1, rs.poisson(cfg.poisson_mean, size=int(target / cfg.poisson_mean * 1.3) + 64)).astype(np.int64)
        cum = np.concatenate([[0], np.cumsum(sizes)])
        n = int(np.searchsorted(cum, target, side="left"))
        cum = cum[:n + 1].copy()
        cum[-1] = target
        _BOUNDS_CACHE[key] = cum
    return _BOUNDS_CACHE[key]


def barcode_name(cfg: SynthConfig, b: int) -> str:
    h = int(_mix(torch.tensor([cfg.seed * 7919 + 13 * b + 5], dtype=torch.int64))[0]) & ((1 << 64) - 1)
    return "".join("ACGT"[(h >> (2 * i)) & 3] for i in range(16)) + f"{b:07d}"


def _chunk_chars(cfg: SynthConfig, p0: int, p1: int, device) -> tuple[torch.Tensor, torch.Tensor]:
    """(code uint8 [n, cpp], valid bool [n, cpp]) for global pairs [p0, p1)"""
    L, cpp = cfg.read_len, cfg.chars_per_pair
    P = cfg.pairs_per_barcode
    p = torch.arange(p0, p1, dtype=torch.int64, device=device) + cfg.first_pair
    local = torch.arange(p0, p1, dtype=torch.int64, device=device)
    bounds = barcode_bounds(cfg)
    if bounds is None:
        barcoded = local < (P * cfg.n_barcodes)
        bc = torch.where(barcoded, p // P, -1 - p)       # unbarcoded pairs behave as their own one-pair barcodes
    else:
        barcoded = local < int(bounds[-1])
        which = torch.searchsorted(torch.from_numpy(bounds[1:]).to(device), local, right=True)
        bc = torch.where(barcoded, which, -1 - p)
    hb = _mix(bc * 0x2545F491 + cfg.seed)
    # genome by inverse CDF of the log-normal abundances
    rs = np.random.RandomState(cfg.seed)
    w = np.exp(rs.randn(cfg.n_genomes))
    cdf = torch.from_numpy(np.cumsum(w / w.sum())).to(device)
    g = torch.searchsorted(cdf, _unit(hb)).clamp_(max=cfg.n_genomes - 1)
    frag_off = (_unit(_mix(hb + 1)) * (cfg.genome_len - cfg.fragment)).to(torch.int64)
    hp = _mix(p * 0x5851F42D + cfg.seed * 31 + 7)
    insert = 300 + (_lsr(hp, 8) % 201)
    pos = frag_off + (_unit(_mix(hp + 3)) * (cfg.fragment - 500)).to(torch.int64)
    flip = (_lsr(hp, 3) & 1).bool()

    j = torch.arange(L, dtype=torch.int64, device=device)[None, :]
    left = pos[:, None] + j                                    # forward read from the left end of the insert
    right = (pos + insert - 1)[:, None] - j                    # reverse-complement read from the right end
    gbase = (g * cfg.genome_len)[:, None]
    idx1 = torch.where(flip[:, None], right, left) + gbase
    idx2 = torch.where(flip[:, None], left, right) + gbase
    rc1 = flip[:, None].expand(-1, L)
    rc2 = ~rc1

    def bases(idx, rc):
        c = _mix(idx * 0x27BB2EE6 + cfg.seed * 131) & 3
        return torch.where(rc, c ^ 2, c)

    code = torch.zeros((p1 - p0, cpp), dtype=torch.int64, device=device)
    valid = torch.zeros((p1 - p0, cpp), dtype=torch.bool, device=device)
    code[:, 0:L] = bases(idx1, rc1)
    code[:, L + 1:2 * L + 1] = bases(idx2, rc2)
    valid[:, 0:L] = True
    valid[:, L + 1:2 * L + 1] = True
    # substitutions: per character
    col = torch.arange(cpp, dtype=torch.int64, device=device)[None, :]
    hc = _mix((p[:, None] * cpp + col) * 0x1B873593 + cfg.seed * 977)
    sub = (_lsr(hc, 20) & 0xFFFFF) < int(cfg.sub_rate * (1 << 20))
    code = torch.where(sub, (code + 1 + (_lsr(hc, 44) % 3)) & 3, code)
    # one N in a fraction of the reads
    for mate, base_col in ((0, 0), (1, L + 1)):
        hn = _mix(p * 2 + mate + cfg.seed * 4099)
        has_n = _unit(hn) < cfg.n_rate
        at = base_col + (_lsr(hn, 5) % L)
        rows = torch.nonzero(has_n).squeeze(1)
        valid[rows, at[rows]] = False
    code = torch.where(valid, code, torch.zeros_like(code))
    return code.to(torch.uint8), valid


def _pack(code: torch.Tensor, valid: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """flat characters (length a multiple of 32) -> (int64 code words, int32 validity words)"""
    c = code.reshape(-1, 32).to(torch.int64)
    v = valid.reshape(-1, 32).to(torch.int64)
    sh = torch.arange(32, dtype=torch.int64, device=code.device)
    cw = (c << (2 * sh)[None, :]).sum(dim=1)
    vw = (v << sh[None, :]).sum(dim=1)
    vw = torch.where(vw >= (1 << 31), vw - (1 << 32), vw).to(torch.int32)
    return cw, vw


def generate(cfg: SynthConfig, device="cpu", chunk_pairs: int = 1 << 16, with_names: bool = True) -> ReadStream:
    device = torch.device(device)
    cpp = cfg.chars_per_pair
    n_chars = cfg.n_pairs * cpp
    n_words = words_for(n_chars)
    codes = torch.zeros(n_words, dtype=torch.int64, device=device)
    valid = torch.zeros(n_words, dtype=torch.int32, device=device)
    chunk_pairs = max(16, chunk_pairs // 16 * 16)            # 16 pairs of 302 characters fill whole words
    if (16 * cpp) % 32:
        raise ValueError("read_len must make 16 pairs a whole number of words (e.g. 150)")
    for p0 in range(0, cfg.n_pairs, chunk_pairs):
        p1 = min(cfg.n_pairs, p0 + chunk_pairs)
        code, ok = _chunk_chars(cfg, p0, p1, device)
        pad = (-(p1 - p0) * cpp) % 32
        code, ok = code.reshape(-1), ok.reshape(-1)
        if pad:
            code = torch.cat([code, code.new_zeros(pad)])
            ok = torch.cat([ok, ok.new_zeros(pad)])
        cw, vw = _pack(code, ok)
        w0 = p0 * cpp // 32
        codes[w0:w0 + cw.numel()] = cw
        valid[w0:w0 + vw.numel()] = vw
    # runs, assembled as the reference does: a run is closed by the first pair of the NEXT barcode, the
    # first pair of the file falls into the leading ""-run, the accumulator left at EOF is one more run
    P, nb = cfg.pairs_per_barcode, cfg.n_barcodes
    bounds = barcode_bounds(cfg)
    if bounds is None:
        n_bc_pairs = min(P * nb, cfg.n_pairs)
        n_bc = (n_bc_pairs + P - 1) // P
        ends = [1] + [min((b + 1) * P + 1, cfg.n_pairs) for b in range(n_bc)]
    else:
        n_bc_pairs, n_bc = int(bounds[-1]), len(bounds) - 1
        ends = [1] + np.minimum(bounds[1:] + 1, cfg.n_pairs).tolist()
    has_tail = n_bc_pairs < cfg.n_pairs
    if not has_tail:
        ends[-1] = cfg.n_pairs
    b0 = cfg.first_pair // P
    names = [""] + [(f"lr_{b:07d}" if bounds is not None else barcode_name(cfg, b0 + b)) if with_names else f"b{b0 + b}" for b in range(n_bc)]
    if has_tail:
        ends.append(cfg.n_pairs)
        names.append("")
    run_off = np.concatenate([[0], np.array(ends, dtype=np.int64)]) * cpp
    return ReadStream(codes, valid, n_chars, run_off.astype(np.int64), names, n_pairs=cfg.n_pairs)


def write_fastq(stream: ReadStream, cfg: SynthConfig, path: str, n_pairs: int | None = None, style: str = "10x") -> int:
    """interleaved FASTQ of the first ``n_pairs`` pairs; returns pairs written.  Header styles:
    ``10x``    ``@r<i> BX:Z:<barcode>-1``                         (run_pangaea -s 10x / tellseq)
    ``stlfr``  ``@r<i>#<b1>_<b2>_<b3>/<mate>``, unbarcoded ``#0_0_0``   (raw stLFR, count_tnf.cpp:35-42)
    ``hybrid`` ``@r<i> BX:Z:lr_<0000000>-1``                      (long-read names as barcodes, assign_barcodes.cpp:156)"""
    n = cfg.n_pairs if n_pairs is None else min(n_pairs, cfg.n_pairs)
    cpp, L, P = cfg.chars_per_pair, cfg.read_len, cfg.pairs_per_barcode
    qual = "I" * L
    bounds = barcode_bounds(cfg)
    with open(path, "w") as f:
        step = 1 << 14
        for p0 in range(0, n, step):
            p1 = min(n, p0 + step)
            txt = stream.decode(p0 * cpp, p1 * cpp).decode()
            out = []
            for p in range(p0, p1):
                o = (p - p0) * cpp
                if bounds is None:
                    b = cfg.first_pair // P + p // P
                    has_bc = p < P * cfg.n_barcodes
                else:
                    b = int(np.searchsorted(bounds[1:], p, side="right"))
                    has_bc = p < int(bounds[-1])
                if style == "stlfr":
                    tag = f"#{b % 1536 + 1}_{(b // 1536) % 1536 + 1}_{b // (1536 * 1536) + 1}" if has_bc else "#0_0_0"
                    h1, h2 = f"@r{p}{tag}/1", f"@r{p}{tag}/2"
                else:
                    name = barcode_name(cfg, b) if (style == "10x" and bounds is None) else f"lr_{b:07d}"
                    h1 = h2 = f"@r{p} BX:Z:{name}-1" if has_bc else f"@r{p}"
                out.append(f"{h1}\n{txt[o:o + L]}\n+\n{qual}\n{h2}\n{txt[o + L + 1:o + 2 * L + 1]}\n+\n{qual}\n")
            f.write("".join(out))
    return n


def write_fastq_fast(stream: ReadStream, cfg: SynthConfig, path: str, n_pairs: int | None = None, chunk_pairs: int = 1 << 17) -> int:
    """``write_fastq(style="10x")`` for files of BASELINE size (10 M pairs = 6.9 GB): the same reads, barcodes and record
    structure, built as fixed-width byte matrices with tensor operations on the stream's device instead of one formatted
    string per pair (minutes -> seconds).  The only difference is the read NAME, zero-padded here
    (``@r00001234 BX:Z:<barcode>-1``): runs, rows and every character of the packed stream come out the same.
    Fixed pairs-per-barcode mode only; returns pairs written."""
    if barcode_bounds(cfg) is not None:
        raise ValueError("write_fastq_fast is for the fixed pairs-per-barcode mode")
    n = cfg.n_pairs if n_pairs is None else min(n_pairs, cfg.n_pairs)
    cpp, L, P = cfg.chars_per_pair, cfg.read_len, cfg.pairs_per_barcode
    dev = stream.device
    first_bc = cfg.first_pair // P
    n_bc_used = max(1, min(cfg.n_barcodes, (n + P - 1) // P))
    # the barcode names of barcode_name(), all at once: 16 bases of a hash + 7 digits
    b = torch.arange(first_bc, first_bc + n_bc_used, dtype=torch.int64, device=dev)
    h = _mix(cfg.seed * 7919 + 13 * b + 5)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    names = torch.empty((n_bc_used, 23), dtype=torch.uint8, device=dev)
    for i in range(16):
        names[:, i] = acgt[(_lsr(h, 2 * i) & 3)]
    for i in range(7):
        names[:, 16 + i] = ((b // 10 ** (6 - i)) % 10 + 48).to(torch.uint8)
    n_barcoded = min(n, P * cfg.n_barcodes)
    lut = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device=dev)          # codes A0 C1 T2 G3
    sh2 = (2 * torch.arange(32, dtype=torch.int64, device=dev))[None, :]
    sh1 = torch.arange(32, dtype=torch.int64, device=dev)[None, :]

    def chars(c0: int, c1: int) -> torch.Tensor:
        w0, w1 = c0 // 32, (c1 + 31) // 32
        c = stream.codes[w0:w1]
        v = stream.valid[w0:w1].to(torch.int64)
        txt = lut[((c[:, None] >> sh2) & 3)]
        txt = torch.where(((v[:, None] >> sh1) & 1).bool(), txt, torch.full_like(txt, ord("N")))
        return txt.reshape(-1)[c0 - 32 * w0:c1 - 32 * w0]

    def block(p0: int, p1: int, barcoded: bool) -> torch.Tensor:
        m = p1 - p0
        txt = chars(p0 * cpp, p1 * cpp).reshape(m, cpp)
        idx = torch.arange(p0, p1, dtype=torch.int64, device=dev)
        w = 42 if barcoded else 11
        hdr = torch.empty((m, w), dtype=torch.uint8, device=dev)
        hdr[:, 0], hdr[:, 1] = ord("@"), ord("r")
        for i in range(8):
            hdr[:, 2 + i] = ((idx // 10 ** (7 - i)) % 10 + 48).to(torch.uint8)
        if barcoded:
            hdr[:, 10:16] = torch.tensor(list(b" BX:Z:"), dtype=torch.uint8, device=dev)
            hdr[:, 16:39] = names[idx // P]
            hdr[:, 39:41] = torch.tensor(list(b"-1"), dtype=torch.uint8, device=dev)
        hdr[:, -1] = ord("\n")
        rec = torch.empty((m, 2 * (w + 2 * L + 4)), dtype=torch.uint8, device=dev)
        o = 0
        for r in range(2):
            rec[:, o:o + w] = hdr; o += w
            rec[:, o:o + L] = txt[:, r * (L + 1):r * (L + 1) + L]; o += L
            rec[:, o:o + 3] = torch.tensor(list(b"\n+\n"), dtype=torch.uint8, device=dev); o += 3
            rec[:, o:o + L] = ord("I"); o += L
            rec[:, o] = ord("\n"); o += 1
        return rec

    with open(path, "wb") as f:
        for p0 in range(0, n_barcoded, chunk_pairs):
            block(p0, min(n_barcoded, p0 + chunk_pairs), True).cpu().numpy().tofile(f)
        for p0 in range(n_barcoded, n, chunk_pairs):
            block(p0, min(n, p0 + chunk_pairs), False).cpu().numpy().tofile(f)
    return n
