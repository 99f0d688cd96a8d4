"""ctypes binding of the C oracle (oracle/_build/libpp_oracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke().
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libpp_oracle.so")
BIN_PATH = os.path.join(HERE, "_build", "pp_oracle")

OK, QUIT, PANIC = 0, 1, 101
STATUS = ("kept", "changed", "low_depth", "none", "multiple", "too_close")


def build(force: bool = False) -> None:
    if force or not (os.path.exists(LIB_PATH) and os.path.exists(BIN_PATH)):
        subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


class OrcError(Exception):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code, self.msg = code, msg


class Buf(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("cap", C.c_size_t)]

    def bytes(self):
        return C.string_at(self.data, self.len) if self.data else b""


class PolishParams(C.Structure):
    _fields_ = [("fraction_invalid", C.c_double), ("fraction_valid", C.c_double),
                ("max_errors", C.c_uint32), ("min_depth", C.c_uint32), ("careful", C.c_int)]


class Positions(C.Structure):
    _fields_ = [("n_positions", C.c_size_t), ("depth", C.POINTER(C.c_double)),
                ("count_a", C.POINTER(C.c_uint32)), ("count_c", C.POINTER(C.c_uint32)),
                ("count_g", C.POINTER(C.c_uint32)), ("count_t", C.POINTER(C.c_uint32)),
                ("count_other", C.POINTER(C.c_uint32)), ("valid_thr", C.POINTER(C.c_uint32)),
                ("invalid_thr", C.POINTER(C.c_uint32)), ("status", C.POINTER(C.c_uint8)),
                ("emit_len", C.POINTER(C.c_uint32))]

    def to_numpy(self):
        n = self.n_positions
        out = {}
        for name, _ in self._fields_[1:]:
            ptr = getattr(self, name)
            out[name] = np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0)
        return out


class PolishCounts(C.Structure):
    _fields_ = [("alignment_total", C.c_uint64), ("used_total", C.c_uint64), ("read_total", C.c_uint64)]


class FilterReport(C.Structure):
    _fields_ = [("before_count", C.c_uint64), ("after_count", C.c_uint64),
                ("low_threshold", C.c_uint32), ("high_threshold", C.c_uint32),
                ("orientation", C.c_int), ("orientation_counts", C.c_uint64 * 4)]


class Records(C.Structure):
    _fields_ = [("n_aln", C.c_uint64), ("contig", C.c_void_p), ("ref_start", C.c_void_p),
                ("k", C.c_void_p), ("seq_off", C.c_void_p), ("seq_len", C.c_void_p),
                ("cig_off", C.c_void_p), ("n_cig", C.c_void_p), ("seq", C.c_void_p),
                ("cigar", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_bankers_rounding.restype = C.c_uint32
        L.orc_bankers_rounding.argtypes = [C.c_double]
        L.orc_reverse_complement.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_get_expanded_cigar.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.orc_get_ref_end.restype = C.c_uint64
        L.orc_get_ref_end.argtypes = [C.c_uint64, C.c_char_p]
        L.orc_get_orientation.argtypes = [C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint32, C.c_uint64, C.c_char_p]
        L.orc_get_insert_size.restype = C.c_uint32
        L.orc_get_insert_size.argtypes = [C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p]
        L.orc_get_percentile.restype = C.c_uint32
        L.orc_get_percentile.argtypes = [C.c_void_p, C.c_size_t, C.c_double]
        L.orc_auto_determine_orientation.argtypes = [C.c_uint64 * 4]
        L.orc_parse_positions.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_pb_new.restype = C.c_void_p
        L.orc_pb_new.argtypes = [C.c_char]
        L.orc_pb_free.argtypes = [C.c_void_p]
        L.orc_pb_add_seq.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_double]
        L.orc_pb_get_polished_seq.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_char_p, C.c_size_t]
        L.orc_pb_get_count_str.argtypes = [C.c_void_p, C.POINTER(Buf)]
        L.orc_read_bases_for_each_target_base.argtypes = [
            C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
            C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.orc_buf_free.argtypes = [C.POINTER(Buf)]
        L.orc_positions_free.argtypes = [C.POINTER(Positions)]
        L.orc_polish_files.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(PolishParams),
                                       C.POINTER(Buf), C.POINTER(Buf), C.POINTER(Positions),
                                       C.POINTER(PolishCounts), C.c_char_p, C.c_size_t]
        L.orc_filter_files.argtypes = [C.c_char_p] * 5 + [C.c_double, C.c_double, C.POINTER(FilterReport),
                                                          C.c_char_p, C.c_size_t]
        L.orc_polish_records.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Records), C.c_uint32,
                                         C.c_double, C.c_double, C.POINTER(Buf), C.c_void_p,
                                         C.POINTER(Positions), C.c_char_p, C.c_size_t]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


# ---- scalar helpers ---------------------------------------------------------
def bankers_rounding(x):
    return lib().orc_bankers_rounding(float(x))


def reverse_complement(s: str) -> str:
    b = s.encode()
    out = C.create_string_buffer(len(b) + 1)
    lib().orc_reverse_complement(b, len(b), out)
    return out.raw[: len(b)].decode()


def get_expanded_cigar(cigar: str):
    p, n = C.c_void_p(), C.c_size_t()
    rc = lib().orc_get_expanded_cigar(cigar.encode(), C.byref(p), C.byref(n))
    if rc != 0:
        return None
    s = C.string_at(p, n.value).decode()
    lib().free(p)
    return s


def get_ref_end(ref_start, cigar):
    return lib().orc_get_ref_end(ref_start, cigar.encode())


def get_orientation(flags1, start1, cigar1, flags2, start2, cigar2):
    return ("fr", "rf", "ff", "rr")[lib().orc_get_orientation(flags1, start1, cigar1.encode(), flags2, start2, cigar2.encode())]


def get_insert_size(start1, cigar1, start2, cigar2):
    return lib().orc_get_insert_size(start1, cigar1.encode(), start2, cigar2.encode())


def get_percentile(sorted_list, p):
    a = np.ascontiguousarray(sorted_list, dtype=np.uint32)
    return lib().orc_get_percentile(a.ctypes.data, len(a), float(p))


def auto_determine_orientation(counts):
    r = lib().orc_auto_determine_orientation((C.c_uint64 * 4)(*counts))
    return None if r < 0 else ("fr", "rf", "ff", "rr")[r]


def parse_positions(line: str):
    s, e = C.c_uint64(), C.c_uint64()
    rc = lib().orc_parse_positions(line.encode(), C.byref(s), C.byref(e))
    if rc != OK:
        raise OrcError(rc, "parse failed")
    return s.value, e.value


class PileupBase:
    def __init__(self, original: str):
        self.h = lib().orc_pb_new(original.encode())

    def add_seq(self, s: str, dc: float):
        b = s.encode()
        lib().orc_pb_add_seq(self.h, b, len(b), dc)

    def get_polished_seq(self, min_depth, fv, fi):
        out = C.create_string_buffer(4096)
        st = lib().orc_pb_get_polished_seq(self.h, min_depth, fv, fi, out, 4096)
        return out.value.decode(), STATUS[st]

    def get_count_str(self):
        b = Buf()
        lib().orc_pb_get_count_str(self.h, C.byref(b))
        s = b.bytes().decode()
        lib().orc_buf_free(C.byref(b))
        return s

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pb_free(self.h)
            self.h = None


def read_slices(cigar: str, seq: str):
    """Strings contributed to consecutive reference positions after the trim ('' = deletion)."""
    b = seq.encode()
    ps, pe, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
    err = C.create_string_buffer(1024)
    rc = lib().orc_read_bases_for_each_target_base(cigar.encode(), b, len(b), C.byref(ps), C.byref(pe),
                                                   C.byref(n), err, 1024)
    if rc != OK:
        raise OrcError(rc, err.value.decode())
    s = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_uint32)), shape=(max(n.value, 1),))[: n.value].copy()
    e = np.ctypeslib.as_array(C.cast(pe, C.POINTER(C.c_uint32)), shape=(max(n.value, 1),))[: n.value].copy()
    lib().free(ps)
    lib().free(pe)
    return [seq[i:j] for i, j in zip(s, e)]


# ---- whole-program drivers ----------------------------------------------------
def polish_files(assembly, sams, fraction_invalid=0.2, fraction_valid=0.5, max_errors=10, min_depth=5,
                 careful=False, debug=False, positions=False):
    L = lib()
    p = PolishParams(fraction_invalid, fraction_valid, max_errors, min_depth, int(careful))
    fasta, dbg, pos, cnt = Buf(), Buf(), Positions(), PolishCounts()
    arr = (C.c_char_p * max(len(sams), 1))(*[s.encode() for s in sams])
    err = C.create_string_buffer(1024)
    rc = L.orc_polish_files(str(assembly).encode(), arr, len(sams), C.byref(p), C.byref(fasta),
                            C.byref(dbg) if debug else None, C.byref(pos) if positions else None,
                            C.byref(cnt), err, 1024)
    try:
        if rc != OK:
            raise OrcError(rc, err.value.decode())
        return {"fasta": fasta.bytes(), "debug": dbg.bytes() if debug else None,
                "positions": pos.to_numpy() if positions else None,
                "counts": (cnt.alignment_total, cnt.used_total, cnt.read_total)}
    finally:
        L.orc_buf_free(C.byref(fasta))
        L.orc_buf_free(C.byref(dbg))
        if positions:
            L.orc_positions_free(C.byref(pos))


def filter_files(in1, in2, out1, out2, orientation="auto", low=0.1, high=99.9):
    rep = FilterReport()
    err = C.create_string_buffer(1024)
    rc = lib().orc_filter_files(str(in1).encode(), str(in2).encode(), str(out1).encode(), str(out2).encode(),
                                orientation.encode(), low, high, C.byref(rep), err, 1024)
    if rc != OK:
        raise OrcError(rc, err.value.decode())
    return {"before": rep.before_count, "after": rep.after_count, "low": rep.low_threshold,
            "high": rep.high_threshold,
            "orientation": ("fr", "rf", "ff", "rr")[rep.orientation] if rep.orientation >= 0 else None,
            "counts": list(rep.orientation_counts)}


def polish_records(contig_off, bases, recs: dict, min_depth=5, fraction_valid=0.5, fraction_invalid=0.2,
                   positions=False):
    """recs: dict of numpy arrays with the C-ABI SoA field names (see include/polypolish_hip.h)."""
    L = lib()
    contig_off = np.ascontiguousarray(contig_off, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    keep = {k: np.ascontiguousarray(recs[k], dtype=dt) for k, dt in (
        ("contig", np.uint32), ("ref_start", np.uint32), ("k", np.uint32), ("seq_off", np.uint64),
        ("seq_len", np.uint32), ("cig_off", np.uint64), ("n_cig", np.uint32), ("seq", np.uint8),
        ("cigar", np.uint32))}
    r = Records(len(keep["contig"]), *[keep[k].ctypes.data for k in
                                       ("contig", "ref_start", "k", "seq_off", "seq_len", "cig_off", "n_cig", "seq", "cigar")])
    n_contigs = len(contig_off) - 1
    out, pos = Buf(), Positions()
    offs = np.zeros(n_contigs + 1, dtype=np.uint64)
    err = C.create_string_buffer(1024)
    rc = L.orc_polish_records(n_contigs, contig_off.ctypes.data, bases.ctypes.data, C.byref(r), min_depth,
                              fraction_valid, fraction_invalid, C.byref(out), offs.ctypes.data,
                              C.byref(pos) if positions else None, err, 1024)
    try:
        if rc != OK:
            raise OrcError(rc, err.value.decode())
        return {"polished": out.bytes(), "offsets": offs, "positions": pos.to_numpy() if positions else None}
    finally:
        L.orc_buf_free(C.byref(out))
        if positions:
            L.orc_positions_free(C.byref(pos))
