"""The arithmetic of k_tile's one-lane-per-read pass over the 4-bit mirror (pp_k_tile.h: wide4_pass, nz_perm, pmask4,
TileShare::pmask, wide4_takes; DESIGN.md "k_tile in detail"), restated in numpy / plain Python and checked against the
obvious per-base loop:
  * nz_perm: bit 4k + d of the result <=> nibble k of dword d differs, i.e. base 8d + k of a 32-base chunk;
  * the range table: pmask[128 + b] = bases [0, clamp(b, 0, 32)) in that bit order, read at 128 - 32c + x for chunk c;
  * a read that starts on an odd base of the seq array is compared through shifted coordinates (nibble x of what is loaded
    <-> window position rel - adj + x, the nibbles [lo + adj, hi + adj) count) instead of being moved down by a nibble;
  * what the pass cannot take (wide4_takes): an odd start whose last base counts when the length fills its last chunk.
(No GPU: this pins the rule, the GPU tests -- test_reads_that_start_on_odd_bases_of_the_seq_array among them -- pin the kernel.)"""
import numpy as np

M32 = 0xFFFFFFFF
PMASK_BASE, NCH, ASM4_PAD, TILE = 128, 5, 32, 2048


def nz_perm(x0, x1, x2, x3):
    m7 = 0x77777777
    u = [(((x & m7) + m7) | x) & M32 for x in (x0, x1, x2, x3)]
    f = (u[0] >> 3) & 0x11111111
    f = ((u[1] >> 2) & 0x22222222) | (f & ~0x22222222 & M32)
    f = ((u[2] >> 1) & 0x44444444) | (f & ~0x44444444 & M32)
    return (u[3] & 0x88888888) | (f & ~0x88888888 & M32)


def pmask4(b):
    m = 0
    for d in range(4):
        n = min(max(b - 8 * d, 0), 8)
        m |= ((0x11111111 if n == 8 else (0x11111111 & ((1 << (4 * n)) - 1))) << d) & M32
    return m


PMASK = [pmask4(min(max(t - PMASK_BASE, 0), 32)) for t in range(PMASK_BASE + 162)]


def test_nz_perm_and_the_range_table():
    rng = np.random.default_rng(5)
    for _ in range(20000):
        x = [int(v) if rng.random() < 0.5 else 0 for v in rng.integers(0, 1 << 32, 4)]
        for d in range(4):  # thin the nibbles out: most bases agree
            for k in range(8):
                if rng.random() < 0.7:
                    x[d] &= ~(0xF << (4 * k)) & M32
        want = {8 * d + k for d in range(4) for k in range(8) if (x[d] >> (4 * k)) & 0xF}
        f = nz_perm(*x)
        got = {8 * (t & 3) + (t >> 2) for t in range(32) if (f >> t) & 1}
        assert got == want
        b0, b1 = sorted(int(v) for v in rng.integers(-PMASK_BASE, 162, 2))  # (the table's whole range: x - 32c, x <= 161, c <= 4)
        f &= PMASK[PMASK_BASE + b1] & ~PMASK[PMASK_BASE + b0] & M32
        got = {8 * (t & 3) + (t >> 2) for t in range(32) if (f >> t) & 1}
        assert got == {i for i in want if b0 <= i < b1}


def wide4_takes(so, notrim, L):
    return not ((so & 1) and notrim and L % 32 == 0)


def pass_model(seq4, asm4_codes, so, L, rel, nkeep):
    """What wide4_pass tallies for one read: the set of (window position, read code) of the differing kept bases.
    seq4: the mirror as an array of nibbles; asm4_codes: the window's codes (position p at index p + ASM4_PAD)."""
    adj = so & 1
    nch = (L + 31) >> 5
    q = so >> 1  # byte of the mirror the loads start at
    lo, hi = max(0, -rel), min(nkeep, TILE - rel)
    if hi <= lo:
        return set()
    relc, xlo, xhi = rel - adj, lo + adj, hi + adj
    out = set()
    for c in range(NCH):
        cc = min(c, nch - 1)                      # a chunk past the last one repeats its address
        base_nib = 2 * (q + 16 * cc)               # nibble index of the chunk's nibble 0 in the mirror
        P0 = relc + 32 * c
        ai = min(max(P0 + ASM4_PAD, 0), 8 * (len(asm4_codes) // 8 - 5))
        W = [0, 0, 0, 0]
        A = [0, 0, 0, 0]
        for x in range(32):
            W[x >> 3] |= int(seq4[base_nib + x]) << (4 * (x & 7))
            A[x >> 3] |= int(asm4_codes[ai + x]) << (4 * (x & 7)) if ai == P0 + ASM4_PAD else 0
        f = nz_perm(*[w ^ a for w, a in zip(W, A)])
        f &= PMASK[PMASK_BASE - 32 * c + xhi] & ~PMASK[PMASK_BASE - 32 * c + xlo] & M32
        for t in range(32):
            if (f >> t) & 1:
                x = 8 * (t & 3) + (t >> 2)
                out.add((P0 + x, (W[t & 3] >> (t & 28)) & 15))
    return out


def test_a_pass_over_reads_at_every_alignment_against_the_per_base_loop():
    rng = np.random.default_rng(9)
    asm = rng.integers(0, 4, TILE + 2 * ASM4_PAD + 64)           # codes A C T G = 0..3; index p + ASM4_PAD
    checked_odd_full = 0
    for trial in range(3000):
        L = int(rng.choice([150, 160, 128, 96, 64, 33, 8, 159, 100]))
        so = int(rng.integers(1, 4000))
        if trial % 3 == 0:
            so |= 1
        notrim = bool(rng.random() < 0.2)
        if not wide4_takes(so, notrim, L):
            continue
        rel = int(rng.integers(-(L - 1), TILE))
        if rng.random() < 0.5:
            rel = int(rng.choice([-(L - 1), -33, -32, -31, -1, 0, 1, TILE - L, TILE - L + 1, TILE - 33, TILE - 1]))
        seq4 = rng.integers(0, 4, 2 * (so // 2 + 16 * NCH) + 64)  # the mirror's nibbles around the read
        read = np.array([asm[ASM4_PAD + rel + i] if 0 <= rel + i < TILE else rng.integers(0, 4) for i in range(L)])
        flip = rng.random(L) < 0.05
        read = np.where(flip, (read + rng.integers(1, 4, L)) % 4, read)
        read[rng.random(L) < 0.01] = 4                          # an N here and there
        seq4[so:so + L] = read
        nkeep = L if notrim else int(rng.integers(0, L - 1))    # a trimmed read never keeps its last base
        want = {(rel + i, int(read[i])) for i in range(max(0, -rel), min(nkeep, TILE - rel)) if read[i] != asm[ASM4_PAD + rel + i]}
        assert pass_model(seq4, asm, so, L, rel, nkeep) == want, (L, so, rel, nkeep, notrim)
        checked_odd_full += (so & 1) and L % 32 == 0
    assert checked_odd_full > 50  # odd starts of lengths that fill their last chunk were among them (trimmed ones)


def test_what_the_pass_hands_on():
    assert wide4_takes(10, True, 160) and wide4_takes(11, False, 160) and wide4_takes(11, True, 150)
    assert not wide4_takes(11, True, 160) and not wide4_takes(3, True, 32)
