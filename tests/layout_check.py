"""Invariants of a batch's seq array as the library's ingests lay it out (include/polypolish_hip.h: PP_SEQ_ALIGN,
PP_SEQ_WINDOW_GROUPED), shared by the CPU and the GPU tests.  Test infrastructure."""
import numpy as np

WINDOW = 2048  # the pileup kernel's window (pp::TILE)


def record_bytes(recs, idx=None):
    """The SEQ bytes of every record (or of records `idx`) as rows of one zero-padded array -- what two batches that lay
    their seq arrays out differently have to agree on."""
    so, sl = recs["seq_off"].astype(np.int64), recs["seq_len"].astype(np.int64)
    if idx is not None:
        so, sl = so[idx], sl[idx]
    if len(sl) == 0:
        return np.zeros((0, 0), dtype=np.uint8)
    width = int(sl.max())
    j = np.arange(width, dtype=np.int64)[None, :]
    live = j < sl[:, None]
    at = np.where(live, so[:, None] + j, 0)
    return np.where(live, recs["seq"][at], 0).astype(np.uint8)


def same_records(a, b):
    """Two batches hold the same records (every array but the placement of the SEQ bytes, and the bytes themselves)."""
    for k in ("contig", "ref_start", "k", "seq_len", "cig_off", "n_cig", "cigar"):
        assert np.array_equal(a[k], b[k]), k
    assert len(a["seq"]) == len(b["seq"])
    assert np.array_equal(record_bytes(a), record_bytes(b)), "SEQ bytes of the records differ"


def check_seq_layout(recs, contig_off, used_per_file, grouped=True, file_order_inside=False):
    """Every record's SEQ on a PP_SEQ_ALIGN boundary, the rooms (SEQ up to the next boundary) tile the seq array exactly, the
    bytes between a read's end and the boundary are zero and no byte of a read is; a file's records take one stretch of the
    array, the stretches in file order.  grouped: inside a file's stretch the windows the records start in come in order
    (file_order_inside: and inside a window the records in file order -- the host ingest; the device tokenizer leaves that
    to its atomics).  Not grouped: the rooms in the order of the records."""
    so, sl = recs["seq_off"].astype(np.int64), recs["seq_len"].astype(np.int64)
    n = len(sl)
    room = (sl + 31) & ~31
    assert (so % 32 == 0).all() and len(recs["seq"]) == int(room.sum())
    order = np.argsort(so, kind="stable")
    assert np.array_equal(so[order], np.cumsum(room[order]) - room[order]), "the rooms do not tile the seq array"
    used = np.zeros(len(recs["seq"]) + 1, dtype=np.int64)
    np.add.at(used, so, 1)
    np.add.at(used, so + sl, -1)
    inside = np.cumsum(used)[:-1]
    assert (recs["seq"][inside == 0] == 0).all() and (recs["seq"][inside == 1] != 0).all()
    if not grouped:
        assert np.array_equal(so, np.cumsum(room) - room), "file order: the rooms follow the records"
        return
    off = np.asarray(contig_off).astype(np.int64)
    n_win = max(1, (int(off[-1]) + WINDOW - 1) // WINDOW)
    win = np.minimum((off[recs["contig"]] + recs["ref_start"].astype(np.int64)) // WINDOW, n_win - 1)
    lo, end_prev = 0, 0
    assert sum(used_per_file) == n
    for cnt in used_per_file:
        hi = lo + cnt
        if cnt:
            assert so[lo:hi].min() == end_prev, "a file's stretch starts where the one before ended"
            end_prev = int((so[lo:hi] + room[lo:hi]).max())
            o = np.argsort(so[lo:hi], kind="stable")
            w = win[lo:hi][o]
            assert (np.diff(w) >= 0).all(), "inside a file's stretch the windows come in order"
            if file_order_inside:
                same = np.diff(w) == 0
                assert (np.diff(o)[same] > 0).all(), "inside a window the records come in file order"
        lo = hi
    assert end_prev == len(recs["seq"])


def check_window_order_mirror(recs, contig_off, used_per_file, file_order_inside=False):
    """pp_aln_batch.wo (include/polypolish_hip.h): a permutation of the batch's records, every entry a copy of its record's
    fields (op0 = the only CIGAR run, or the marker for a record of several), a file's records in one stretch, inside it the
    windows the records start in in order (file_order_inside: and inside a window the records in file order -- the host
    ingest; the device tokenizer leaves that to its atomics)."""
    wo = recs["wo"]
    n = len(recs["contig"])
    assert len(wo) == n
    fi = wo["file_idx"].astype(np.int64)
    assert np.array_equal(np.sort(fi), np.arange(n)), "not a permutation of the records"
    for k in ("contig", "ref_start", "k", "seq_len", "seq_off"):
        assert np.array_equal(wo[k], recs[k][fi]), k
    one = recs["n_cig"][fi] == 1
    assert (wo["op0"][~one] == 0xFFFFFFFF).all()
    assert np.array_equal(wo["op0"][one], recs["cigar"][recs["cig_off"][fi][one].astype(np.int64)])
    off = np.asarray(contig_off).astype(np.int64)
    n_win = max(1, (int(off[-1]) + WINDOW - 1) // WINDOW)
    ctg = np.minimum(wo["contig"].astype(np.int64), len(off) - 2)   # (a record that names no contig of the assembly is still a record)
    win = np.minimum((off[ctg] + wo["ref_start"].astype(np.int64)) // WINDOW, n_win - 1)
    lo = 0
    for cnt in used_per_file:
        hi = lo + cnt
        assert ((fi[lo:hi] >= lo) & (fi[lo:hi] < hi)).all(), "a file's records take one stretch of the mirror"
        assert (np.diff(win[lo:hi]) >= 0).all(), "inside a file's stretch the windows come in order"
        if file_order_inside:
            same = np.diff(win[lo:hi]) == 0
            assert (np.diff(fi[lo:hi])[same] > 0).all(), "inside a window the records come in file order"
        lo = hi
    assert lo == n
    # the mirror's run table (pp_aln_batch.wo_run_end): one run per file that yielded records, ends ascending, the last one n
    if "wo_runs" in recs:
        want = [int(e) for e, c in zip(np.cumsum(used_per_file), used_per_file) if c]
        assert [int(e) for e in recs["wo_runs"]] == want, (recs["wo_runs"], want)
