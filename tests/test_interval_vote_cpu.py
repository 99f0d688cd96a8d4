"""The arithmetic behind k_tile's interval-bound vote (pp_k_tile.h): a position's depth is the reference's
ordered f64 sum of 1.0/k (src/pileup.rs:64, src/alignment.rs:288); the kernel tallies the shares in fixed point, 2^-b per
unit, every share rounded to the nearest unit, and votes from the interval [D - n*2^-(b+1) - 1e-9, D + ...] whenever the
vote's three step functions of the depth (src/pileup.rs:70-72,114) take the same values at both ends.  Here in numpy, on
random mixes of k: whenever the interval test says "decided", the thresholds and the depth test must be the ones the exact
ordered sum gives -- for every order of the reads.  (No GPU: this pins the rule, the GPU tests pin the kernel.)"""
import numpy as np


def bankers(x):  # src/misc.rs:208-215 for x >= 0
    r = np.floor(x).astype(np.int64)
    f = x - np.floor(x)
    return np.where(f < 0.5, r, np.where(f > 0.5, r + 1, r + (r & 1)))


def share_units(k, b):
    one = 1 << b
    return (one + (k >> 1)) // k


def test_share_rounding_is_within_half_a_unit():
    for b in (10, 13, 19, 20):
        k = np.arange(1, 5000, dtype=np.int64)
        s = share_units(k, b)
        err = np.abs(s / float(1 << b) - 1.0 / k)
        assert (err <= 0.5 / (1 << b) + 1e-18).all()
        exact = (s * k == (1 << b))
        assert exact[(k & (k - 1)) == 0][: b + 1].all()          # powers of two up to 2^b are exact
        assert not exact[(k & (k - 1)) != 0].any()               # nothing else is


def test_decided_positions_get_the_exact_thresholds():
    rng = np.random.default_rng(11)
    n_pos, undecided = 0, 0
    for trial in range(400):
        b = int(rng.choice([10, 12, 19, 20]))
        n = int(rng.integers(1, 400))
        ks = rng.choice([1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 12, 1000, 1024, 3000], size=n).astype(np.int64)
        fv, fi = float(rng.choice([0.5, 0.6, 0.37])), float(rng.choice([0.2, 0.05, 0.11]))
        min_depth = int(rng.choice([0, 1, 5, 40]))
        # the kernel's fixed-point depth and interval
        deficit = int(((1 << b) - share_units(ks, b)).sum())
        D = ((n << b) - deficit) / float(1 << b)
        eps = n * (0.5 / (1 << b)) + 1e-9
        lo, hi = max(D - eps, 0.0), D + eps
        t = [(int(bankers(np.float64(x) * fv)), int(bankers(np.float64(x) * fi)), x < min_depth) for x in (lo, hi)]
        decided = t[0] == t[1]
        n_pos += 1
        undecided += not decided
        for order in range(3):  # the reference's sum in three different file orders
            perm = rng.permutation(n)
            depth = 0.0
            for k in ks[perm]:
                depth += 1.0 / float(k)
            assert lo <= depth <= hi, (b, n, D, depth)
            exact = (int(bankers(np.float64(depth) * fv)), int(bankers(np.float64(depth) * fi)), depth < min_depth)
            if decided:
                assert exact == t[0]
            elif sum(a != c for a, c in zip(t[0], t[1])) == 1 and t[1][0] - t[0][0] <= 1 and t[1][1] - t[0][1] <= 1:
                # one of the three moved by one step between the ends: the exact depth gives one end's values or the other's
                # (the kernel then votes with both and only replays the position if the two votes differ)
                assert exact in (t[0], t[1]), (exact, t)
    assert undecided < 0.25 * n_pos   # (b = 10 with hundreds of inexact reads leaves a wide interval; b >= 19 almost none)


def win_fx_bits(n_items):  # pp_k_common.h: the unit 2^-b of a window with n_items work items
    return min(20, 31 - max(int(n_items), 1).bit_length())


def test_interval_holds_the_ordered_sum_of_a_deep_window_of_odd_shares():
    """The systematic case: EVERY read with the same inexact share, 10^5 and more of them on one position --
    the share's rounding error has one sign and adds up (k = 3: each unit a third too small at b = 14), and the ordered
    f64 sum drifts on its own.  The kernel's interval, n * 2^-(b+1) + 1e-9 either side of the fixed-point depth with b as
    it picks it for a window of n items, must still hold the reference's sum, in file order and in any other."""
    rng = np.random.default_rng(5)
    for n, k in ((100_000, 3), (250_000, 3), (131_071, 7), (200_000, 6), (1_000_000, 3)):
        b = win_fx_bits(n)
        deficit = n * ((1 << b) - share_units(np.int64(k), b))
        D = ((n << b) - int(deficit)) / float(1 << b)
        eps = n * (0.5 / (1 << b)) + 1e-9
        depth = float(np.add.accumulate(np.full(n, 1.0 / k))[-1])   # (sequential f64 additions, as src/pileup.rs:64)
        assert D - eps <= depth <= D + eps, (n, k, b, D, depth, eps)
        assert abs(depth - n / k) < 1e-4 < eps   # (the f64 sum's own drift -- 1e-6 at 10^6 reads -- is far inside the interval's half-width)
    # ... and mixed with exact shares in a random order
    n = 150_000
    ks = rng.choice([1, 2, 3, 3, 3, 5], size=n).astype(np.int64)
    b = win_fx_bits(n)
    D = ((n << b) - int(((1 << b) - share_units(ks, b)).sum())) / float(1 << b)
    eps = n * (0.5 / (1 << b)) + 1e-9
    depth = float(np.add.accumulate(1.0 / ks.astype(np.float64))[-1])
    assert D - eps <= depth <= D + eps
