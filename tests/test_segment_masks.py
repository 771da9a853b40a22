"""The record ends of the first scatter pass as bit arithmetic (mini.hip: mini_record_ends) against the position-by-position
form it replaced (a counter, four comparisons and a predicated block per position): Python restatements of both, on random masks.
The device function itself is covered by every table-against-oracle test of the GPU suite; this pins the ALGEBRA -- including
the cap on the record length, which the mask form expresses as "no run of `cap` continuation bits"."""
import random

M = 0xFFFFFFFF


def runs32(m, n):
    r, ln = m, 1
    while 2 * ln <= n:
        r &= (r << ln) & M
        ln *= 2
    if ln < n:
        r &= (r << (n - ln)) & M
    return r


def ends_by_masks(ok, ok_row, cuts, eq, cap):
    cont = ok & ((ok << 1) & M) & ~(ok_row ^ ((ok_row << 1) & M)) & ~cuts & eq & M
    if cap <= 1:
        cont = 0
    else:
        over = runs32(cont, cap)
        while over:
            cont &= ~(over & -over) & M
            over = runs32(cont, cap)
    return ok & ~(cont >> 1) & M


def ends_by_positions(ok, ok_row, cuts, values, cap):
    same = ok & ((ok << 1) & M) & ~(ok_row ^ ((ok_row << 1) & M)) & ~cuts & M
    n, cur, has = 0, 0, 0
    for p in range(32):
        mv, v = values[p], (ok >> p) & 1
        cont = bool((same >> p) & 1) and n > 0 and mv == cur and n < cap
        if p > 0 and n > 0 and not cont:
            has |= 1 << (p - 1)
        n = n + 1 if cont else (1 if v else 0)
        cur = mv
    if n > 0:
        has |= 1 << 31
    return has


def test_mask_form_equals_the_position_by_position_form():
    rnd = random.Random(1)
    for _ in range(60_000):
        ok = rnd.getrandbits(32) | rnd.getrandbits(32) | rnd.getrandbits(32)
        ok_row = ok & (rnd.getrandbits(32) | rnd.getrandbits(32)) if rnd.random() < 0.5 else ok
        cuts = rnd.getrandbits(32) & rnd.getrandbits(32) & rnd.getrandbits(32) & rnd.getrandbits(32)
        values, cur = [], rnd.getrandbits(8)
        for _p in range(32):
            if rnd.random() < 0.25:
                cur = rnd.getrandbits(8)
            values.append(cur)
        eq, prev = 0, 0
        for p in range(32):
            if values[p] == prev:
                eq |= 1 << p
            prev = values[p]
        cap = rnd.choice([1, 2, 3, 4, 6, 9, 16])
        assert ends_by_masks(ok, ok_row, cuts, eq, cap) == ends_by_positions(ok, ok_row, cuts, values, cap)
