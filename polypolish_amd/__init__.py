"""polypolish_amd -- Python face of libpolypolish_hip.so (MI355X / gfx950).

The product is the C-ABI library (include/polypolish_hip.h) and the ``bin/polypolish`` CLI; this
package is a thin ctypes binding used by the tests, bench.py and anyone who wants to call the
hot path from Python.  It mirrors the reference's two drivers by name and argument meaning:

    polish(assembly, sam, ...)            <->  polish::polish  (src/polish.rs:26-38)
    filter(in1, in2, out1, out2, ...)     <->  filter::filter  (src/filter.rs:26-37)

There is no CPU fallback: importing works anywhere the shared library loads, but every compute
call needs a HIP device and raises ``PolypolishError`` otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("PP_LIB_PATH") or os.path.join(HERE, "_build", "libpolypolish_hip.so")  # PP_LIB_PATH: kernel experiments

OK, ERR_QUIT, ERR_HIP, ERR_ARG, ERR_LIMIT, ERR_PANIC = 0, 1, 3, 4, 5, 101
ERR_NOT_ASCII = 6  # device text front ends only: bytes outside ASCII, the host parsers take such a file
MEM_HOST, MEM_DEVICE, MEM_PEER = 0, 1, 2
STATUS = ("kept", "changed", "low_depth", "none", "multiple", "too_close")
OPS = "MIDNSHP=X"


class PolypolishError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code, self.msg = code, msg


class Params(C.Structure):
    _fields_ = [("min_depth", C.c_uint32), ("fraction_valid", C.c_double), ("fraction_invalid", C.c_double)]


class AlnBatch(C.Structure):
    _fields_ = [("n_aln", C.c_uint64), ("contig", C.c_void_p), ("ref_start", C.c_void_p), ("k", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq_len", C.c_void_p), ("cig_off", C.c_void_p), ("n_cig", C.c_void_p),
                ("seq", C.c_void_p), ("seq_bytes", C.c_uint64), ("cigar", C.c_void_p), ("n_cig_total", C.c_uint64),
                ("seq4", C.c_void_p),  # optional 4-bit mirror of seq (include/polypolish_hip.h); None = none
                ("wo", C.c_void_p),    # optional window-order mirror of the records (pp_wo_rec[n_aln]); None = none
                ("wo_n_runs", C.c_uint32), ("wo_run_end", C.c_void_p)]  # optional: the mirror's runs (HOST array of u64 ends)


def _runs_of(b):
    """the run table of a batch's window-order mirror (pp_aln_batch.wo_run_end: HOST memory) as a numpy array"""
    if not b.wo_n_runs or not b.wo_run_end:
        return np.zeros(0, dtype=np.uint64)
    return np.ctypeslib.as_array(C.cast(b.wo_run_end, C.POINTER(C.c_uint64)), shape=(int(b.wo_n_runs),)).copy()


def aln_batch(n_aln, ptrs: dict, seq_bytes, n_cig_total):
    """pp_aln_batch from addresses (field name -> address; "seq4", "wo" optional) plus, optionally, ptrs["wo_runs"]: the ends
    of the mirror's runs (a sequence of integers, HOST: pp_aln_batch.wo_run_end)."""
    runs = ptrs.get("wo_runs")
    if runs is not None:
        runs = np.ascontiguousarray(runs, dtype=np.uint64)
    b = AlnBatch(n_aln, ptrs["contig"], ptrs["ref_start"], ptrs["k"], ptrs["seq_off"], ptrs["seq_len"],
                 ptrs["cig_off"], ptrs["n_cig"], ptrs["seq"], seq_bytes, ptrs["cigar"], n_cig_total, ptrs.get("seq4") or None,
                 ptrs.get("wo") or None, len(runs) if runs is not None and ptrs.get("wo") else 0,
                 runs.ctypes.data if runs is not None and len(runs) and ptrs.get("wo") else None)
    b._runs = runs  # (the C side reads it during pp_polish_add)
    return b


# one record of pp_aln_batch.wo (include/polypolish_hip.h: pp_wo_rec, 32 bytes)
WO_DTYPE = np.dtype([("contig", np.uint32), ("ref_start", np.uint32), ("k", np.uint32), ("seq_len", np.uint32),
                     ("seq_off", np.uint64), ("op0", np.uint32), ("file_idx", np.uint32)])
WO_MULTI_RUN = 0xFFFFFFFF


def window_order_mirror(recs, contig_off, used_per_file=None, window=2048):
    """pp_aln_batch.wo for a batch given as numpy arrays: its records in window order -- per SAM file (used_per_file: good
    records of every file; None = one file), the records that start in one 2048-position window adjacent, file order inside
    a window -- as the host ingest writes it."""
    n = len(recs["contig"])
    off = np.asarray(contig_off).astype(np.int64)
    n_win = max(1, (int(off[-1]) + window - 1) // window)
    c = np.minimum(recs["contig"].astype(np.int64), len(off) - 2)
    win = np.minimum((off[c] + recs["ref_start"].astype(np.int64)) // window, n_win - 1)
    file_of = np.zeros(n, dtype=np.int64)
    if used_per_file is not None:
        file_of = np.repeat(np.arange(len(used_per_file)), used_per_file)
    order = np.argsort(file_of * n_win + win, kind="stable")
    wo = np.zeros(n, dtype=WO_DTYPE)
    for k in ("contig", "ref_start", "k", "seq_len", "seq_off"):
        wo[k] = recs[k][order]
    first = recs["cigar"][np.minimum(recs["cig_off"][order].astype(np.int64), max(len(recs["cigar"]) - 1, 0))] if len(recs["cigar"]) else np.zeros(n, np.uint32)
    wo["op0"] = np.where(recs["n_cig"][order] == 1, first, WO_MULTI_RUN)
    wo["file_idx"] = order.astype(np.uint32)
    return wo


class ContigStats(C.Structure):
    _fields_ = [("polished_len", C.c_uint64), ("changed", C.c_uint64), ("zero_depth", C.c_uint64),
                ("depth_sum", C.c_double)]


class PositionsOut(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("count_a", C.c_void_p), ("count_c", C.c_void_p), ("count_g", C.c_void_p),
                ("count_t", C.c_void_p), ("count_other", C.c_void_p), ("valid_thr", C.c_void_p),
                ("invalid_thr", C.c_void_p), ("status", C.c_void_p)]


PP_MAX_KERNELS = 16


class KernelTimes(C.Structure):
    _fields_ = [("n", C.c_int), ("name", C.c_char_p * PP_MAX_KERNELS), ("ms", C.c_float * PP_MAX_KERNELS),
                ("n_entries", C.c_uint64), ("n_flagged", C.c_uint64), ("n_passes", C.c_uint64)]

    def as_dict(self):
        d = {self.name[i].decode(): float(self.ms[i]) for i in range(self.n)}
        return {"ms": d, "n_entries": int(self.n_entries), "n_flagged": int(self.n_flagged),
                "n_passes": int(self.n_passes)}


class SamCounts(C.Structure):
    _fields_ = [("alignments", C.c_uint64), ("used", C.c_uint64), ("reads", C.c_uint64)]


class PolishOptions(C.Structure):
    _fields_ = [("fraction_invalid", C.c_double), ("fraction_valid", C.c_double), ("max_errors", C.c_uint32),
                ("min_depth", C.c_uint32), ("careful", C.c_int), ("debug_path", C.c_char_p), ("quiet", C.c_int)]


class Bytes(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_uint64)]


class FilterReport(C.Structure):
    _fields_ = [("before_count", C.c_uint64), ("after_count", C.c_uint64), ("low_threshold", C.c_uint32),
                ("high_threshold", C.c_uint32), ("orientation", C.c_int), ("orientation_counts", C.c_uint64 * 4)]


class FilterFile(C.Structure):
    _fields_ = [("n_aln", C.c_uint64), ("ref_id", C.c_void_p), ("ref_start", C.c_void_p), ("flags", C.c_void_p),
                ("cig_off", C.c_void_p), ("n_cig", C.c_void_p), ("cigar", C.c_void_p), ("n_cig_total", C.c_uint64),
                ("read", C.c_void_p), ("grp_off", C.c_void_p), ("grp_idx", C.c_void_p), ("ref_end", C.c_void_p)]


class FilterInput(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("file", FilterFile * 2)]


class ShardPlan(C.Structure):
    _fields_ = [("n_units", C.c_uint32), ("world", C.c_uint32), ("n_contigs", C.c_uint32), ("contig", C.POINTER(C.c_uint32)),
                ("lo", C.POINTER(C.c_uint64)), ("hi", C.POINTER(C.c_uint64)), ("rank", C.POINTER(C.c_uint32))]


COMM_ID_BYTES = 128


class FilterFileCounts(C.Structure):
    _fields_ = [("alignments", C.c_uint64), ("reads", C.c_uint64), ("loaded", C.c_int)]


# every symbol include/polypolish_hip.h declares (tests check that the library exports them all)
EXPORTS = [
    "pp_device_count", "pp_ctx_create", "pp_ctx_create_async", "pp_ctx_wait", "pp_ctx_destroy", "pp_last_error", "pp_ctx_sync", "pp_ctx_stream", "pp_ctx_download", "pp_version", "pp_log_text",
    "pp_polish_begin", "pp_polish_add", "pp_polish_reserve", "pp_polish_finish", "pp_polish_result_size", "pp_polish_result",
    "pp_polish_result_device", "pp_polish_set_emit", "pp_polish_set_debug", "pp_polish_positions", "pp_polish_debug_extra",
    "pp_debug_extra_free", "pp_ctx_set_profiling",
    "pp_polish_kernel_times", "pp_polish_took_direct_path", "pp_filter_begin", "pp_filter_samples", "pp_filter_pairs",
    "pp_filter_kernel_times", "pp_filter_load", "pp_filter_loaded_input", "pp_filter_write", "pp_filter_loaded_free",
    "pp_filter_load_device", "pp_filter_dev_input", "pp_filter_dev_text", "pp_filter_dev_free", "pp_filter_write_text",
    "pp_assembly_load", "pp_assembly_free", "pp_assembly_n_contigs",
    "pp_assembly_name", "pp_assembly_description", "pp_assembly_offsets", "pp_assembly_bases",
    "pp_ingest_create", "pp_ingest_sam", "pp_ingest_batch", "pp_ingest_read_name", "pp_ingest_free",
    "pp_bytes_free", "pp_polish_files", "pp_filter_files", "pp_filter_polish_files", "pp_ingest_sam_filtered",
    "pp_dev_ingest_create", "pp_dev_ingest_sam", "pp_dev_ingest_sam_filtered", "pp_dev_ingest_batch", "pp_dev_ingest_free",
    "pp_shard_plan_create", "pp_shard_plan_free", "pp_shard_emit_ranges", "pp_shard_assemble",
    "pp_comm_unique_id", "pp_comm_init", "pp_comm_destroy", "pp_polish_gather", "pp_polish_files_multi",
    "pp_shard_split", "pp_shard_part_batch", "pp_shard_part_mem", "pp_shard_part_free", "pp_shard_count",
    "pp_polish_error_record", "pp_polish_error_text", "pp_dev_ingest_set_seq_layout", "pp_dev_ingest_expect",
    "pp_ingest_set_seq_layout",
]

_lib = None


def lib():
    """Load the HIP library; fails loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `make` (or __graft_entry__.build()) first; "
                              "polypolish_amd has no pure-Python or CPU path")
        L = C.CDLL(LIB_PATH)
        vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
        L.pp_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.pp_ctx_create_async.argtypes = [C.c_int, C.POINTER(vp)]
        L.pp_ctx_wait.argtypes = [vp]
        L.pp_ctx_destroy.argtypes = [vp]
        L.pp_ctx_destroy.restype = None
        L.pp_last_error.argtypes = [vp]
        L.pp_last_error.restype = C.c_char_p
        L.pp_ctx_sync.argtypes = [vp]
        L.pp_ctx_download.argtypes = [vp, vp, vp, C.c_uint64]
        L.pp_ctx_stream.argtypes = [vp]
        L.pp_ctx_stream.restype = vp
        L.pp_version.restype = C.c_char_p
        L.pp_log_text.argtypes = [C.c_int, C.c_double, C.c_char_p, C.c_size_t]
        L.pp_polish_begin.argtypes = [vp, C.c_uint32, vp, vp, C.c_int, C.POINTER(Params)]
        L.pp_polish_add.argtypes = [vp, C.POINTER(AlnBatch), C.c_int]
        L.pp_polish_finish.argtypes = [vp]
        L.pp_polish_result_size.argtypes = [vp, u64p]
        L.pp_polish_result.argtypes = [vp, vp, C.c_int, vp, vp]
        L.pp_polish_result_device.argtypes = [vp]
        L.pp_polish_result_device.restype = vp
        L.pp_polish_set_debug.argtypes = [vp, C.c_int]
        L.pp_polish_set_emit.argtypes = [vp, vp, vp]
        L.pp_polish_positions.argtypes = [vp, C.POINTER(PositionsOut)]
        L.pp_ctx_set_profiling.argtypes = [vp, C.c_int]
        L.pp_polish_kernel_times.argtypes = [vp, C.POINTER(KernelTimes)]
        L.pp_polish_took_direct_path.argtypes = [vp]
        L.pp_filter_begin.argtypes = [vp, C.POINTER(FilterInput), C.c_int]
        L.pp_filter_samples.argtypes = [vp, vp, vp]
        L.pp_filter_pairs.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint8, vp, vp]
        L.pp_filter_kernel_times.argtypes = [vp, C.POINTER(KernelTimes)]
        L.pp_filter_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp), C.POINTER(FilterFileCounts), C.c_char_p, C.c_size_t]
        L.pp_filter_loaded_input.argtypes = [vp, C.POINTER(FilterInput)]
        L.pp_filter_loaded_input.restype = None
        L.pp_filter_write.argtypes = [vp, C.c_int, vp, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                      C.c_char_p, C.c_size_t]
        L.pp_filter_loaded_free.argtypes = [vp]
        L.pp_filter_loaded_free.restype = None
        L.pp_filter_load_device.argtypes = [vp, C.c_char_p, C.c_char_p, C.POINTER(vp), C.POINTER(FilterFileCounts)]
        L.pp_filter_dev_input.argtypes = [vp, C.POINTER(FilterInput)]
        L.pp_filter_dev_input.restype = None
        L.pp_filter_dev_text.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
        L.pp_filter_dev_text.restype = vp
        L.pp_filter_dev_free.argtypes = [vp]
        L.pp_filter_dev_free.restype = None
        L.pp_filter_write_text.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                           C.c_char_p, C.c_size_t]
        L.pp_assembly_load.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]
        L.pp_assembly_free.argtypes = [vp]
        L.pp_assembly_free.restype = None
        L.pp_assembly_n_contigs.argtypes = [vp]
        L.pp_assembly_n_contigs.restype = C.c_uint32
        L.pp_assembly_name.argtypes = [vp, C.c_uint32]
        L.pp_assembly_name.restype = C.c_char_p
        L.pp_assembly_description.argtypes = [vp, C.c_uint32]
        L.pp_assembly_description.restype = C.c_char_p
        L.pp_assembly_offsets.argtypes = [vp]
        L.pp_assembly_offsets.restype = u64p
        L.pp_assembly_bases.argtypes = [vp]
        L.pp_assembly_bases.restype = C.POINTER(C.c_uint8)
        L.pp_ingest_create.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(vp)]
        L.pp_ingest_sam.argtypes = [vp, C.c_char_p, C.POINTER(SamCounts), C.c_char_p, C.c_size_t]
        L.pp_ingest_batch.argtypes = [vp, C.POINTER(AlnBatch)]
        L.pp_ingest_batch.restype = None
        L.pp_ingest_read_name.argtypes = [vp, C.c_uint64]
        L.pp_ingest_read_name.restype = C.c_char_p
        L.pp_ingest_free.argtypes = [vp]
        L.pp_ingest_free.restype = None
        L.pp_bytes_free.argtypes = [C.POINTER(Bytes)]
        L.pp_bytes_free.restype = None
        L.pp_polish_files.argtypes = [vp, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(PolishOptions),
                                      C.POINTER(Bytes)]
        L.pp_filter_files.argtypes = [vp] + [C.c_char_p] * 5 + [C.c_double, C.c_double, C.c_int,
                                                                C.POINTER(FilterReport)]
        L.pp_filter_polish_files.argtypes = [vp] + [C.c_char_p] * 6 + [C.c_double, C.c_double, C.POINTER(PolishOptions),
                                                                       C.POINTER(FilterReport), C.POINTER(Bytes)]
        L.pp_dev_ingest_create.argtypes = [vp, vp, C.c_uint32, C.c_int, C.POINTER(vp)]
        L.pp_dev_ingest_sam.argtypes = [vp, C.c_char_p, C.POINTER(SamCounts)]
        L.pp_dev_ingest_sam_filtered.argtypes = [vp, C.c_char_p, vp, C.c_uint64, C.POINTER(SamCounts)]
        L.pp_dev_ingest_set_seq_layout.argtypes = [vp, C.c_int]
        L.pp_ingest_set_seq_layout.argtypes = [vp, C.c_int]
        L.pp_dev_ingest_batch.argtypes = [vp, C.POINTER(AlnBatch)]
        L.pp_dev_ingest_batch.restype = None
        L.pp_dev_ingest_free.argtypes = [vp]
        L.pp_dev_ingest_free.restype = None
        L.pp_ingest_sam_filtered.argtypes = [vp, C.c_char_p, vp, C.c_uint64, C.POINTER(SamCounts), C.c_char_p, C.c_size_t]
        L.pp_shard_plan_create.argtypes = [C.c_uint32, vp, vp, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(ShardPlan))]
        L.pp_shard_plan_free.argtypes = [C.POINTER(ShardPlan)]
        L.pp_shard_plan_free.restype = None
        L.pp_shard_emit_ranges.argtypes = [C.POINTER(ShardPlan), C.c_uint32, vp, vp]
        L.pp_shard_assemble.argtypes = [C.POINTER(ShardPlan), vp, vp, vp, vp]
        L.pp_comm_unique_id.argtypes = [vp]
        L.pp_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
        L.pp_comm_destroy.argtypes = [vp]
        L.pp_comm_destroy.restype = None
        L.pp_polish_gather.argtypes = [vp, vp, C.c_uint64, vp, vp]
        L.pp_shard_split.argtypes = [vp, C.POINTER(ShardPlan), C.c_uint32, C.POINTER(AlnBatch), C.c_int, C.POINTER(vp)]
        L.pp_shard_part_batch.argtypes = [vp, C.POINTER(AlnBatch), C.POINTER(vp)]
        L.pp_shard_part_batch.restype = None
        L.pp_shard_part_mem.argtypes = [vp]
        L.pp_shard_part_free.argtypes = [vp]
        L.pp_shard_part_free.restype = None
        L.pp_shard_count.argtypes = [vp, C.POINTER(AlnBatch), C.c_int, C.c_uint32, vp]
        L.pp_polish_error_record.argtypes = [vp, u64p, C.POINTER(C.c_uint32)]
        L.pp_polish_error_text.argtypes = [vp, C.c_uint32, C.c_uint64]
        _lib = L
    return _lib


REC_FIELDS = (("contig", np.uint32), ("ref_start", np.uint32), ("k", np.uint32), ("seq_off", np.uint64),
              ("seq_len", np.uint32), ("cig_off", np.uint64), ("n_cig", np.uint32), ("seq", np.uint8),
              ("cigar", np.uint32))


def split_records(recs, cuts):
    """The records cut at the indices `cuts` into batches that are valid on their own: seq_off / cig_off relative to
    each batch's seq / cigar arrays (which hold exactly the bytes / runs its records refer to, re-packed)."""
    n = len(recs["contig"])
    edges = [0] + sorted(int(c) for c in (cuts or []) if 0 < int(c) < n) + [n]
    if len(edges) == 2:
        return [recs]
    out = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        sl = recs["seq_len"][lo:hi].astype(np.int64)
        so = recs["seq_off"][lo:hi].astype(np.int64)
        nc = recs["n_cig"][lo:hi].astype(np.int64)
        co = recs["cig_off"][lo:hi].astype(np.int64)
        new_so = np.concatenate([[0], np.cumsum(sl)[:-1]]) if hi > lo else np.zeros(0, np.int64)
        new_co = np.concatenate([[0], np.cumsum(nc)[:-1]]) if hi > lo else np.zeros(0, np.int64)
        seq_idx = np.repeat(so - new_so, sl) + np.arange(int(sl.sum())) if hi > lo else np.zeros(0, np.int64)
        cig_idx = np.repeat(co - new_co, nc) + np.arange(int(nc.sum())) if hi > lo else np.zeros(0, np.int64)
        part = {k: np.ascontiguousarray(recs[k][lo:hi]) for k in ("contig", "ref_start", "k", "seq_len", "n_cig")}
        part["seq_off"] = new_so.astype(np.uint64)
        part["cig_off"] = new_co.astype(np.uint64)
        part["seq"] = np.ascontiguousarray(recs["seq"][seq_idx]) if len(seq_idx) else np.zeros(0, np.uint8)
        part["cigar"] = np.ascontiguousarray(recs["cigar"][cig_idx]) if len(cig_idx) else np.zeros(0, np.uint32)
        out.append(part)
    return out


def ingest(assembly, sams, max_errors=10, careful=False, seq_layout=None):
    """Host ingest only (no GPU needed): FASTA + SAM text -> (names, descs, contig_off, bases, recs, counts).
    seq_layout: None = the library's default (window-grouped unless PP_SEQ_LAYOUT=file), 0 = SEQ bytes in file order,
    1 = window-grouped (pp_ingest_set_seq_layout)."""
    L = lib()
    err = C.create_string_buffer(1024)
    a = C.c_void_p()
    rc = L.pp_assembly_load(str(assembly).encode(), C.byref(a), err, 1024)
    if rc:
        raise PolypolishError(rc, err.value.decode())
    g = C.c_void_p()
    try:
        n = L.pp_assembly_n_contigs(a)
        names = [L.pp_assembly_name(a, i).decode() for i in range(n)]
        descs = [L.pp_assembly_description(a, i).decode() for i in range(n)]
        off = np.ctypeslib.as_array(L.pp_assembly_offsets(a), shape=(n + 1,)).copy()
        bases = np.ctypeslib.as_array(L.pp_assembly_bases(a), shape=(int(off[-1]),)).copy()
        L.pp_ingest_create(a, max_errors, int(careful), C.byref(g))
        if seq_layout is not None:
            L.pp_ingest_set_seq_layout(g, int(seq_layout))
        counts = []
        for s in sams:
            c = SamCounts()
            rc = L.pp_ingest_sam(g, str(s).encode(), C.byref(c), err, 1024)
            if rc:
                raise PolypolishError(rc, err.value.decode())
            counts.append((c.alignments, c.used, c.reads))
        b = AlnBatch()
        L.pp_ingest_batch(g, C.byref(b))
        sizes = {"seq": b.seq_bytes, "cigar": b.n_cig_total}
        recs = {}
        for name, dt in REC_FIELDS:
            cnt = int(sizes.get(name, b.n_aln))
            ptr = getattr(b, name)
            if cnt and ptr:
                arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()
            else:
                arr = np.zeros(0, dtype=dt)
            recs[name] = arr
        if b.wo and b.n_aln:  # the window-order mirror of the records (pp_aln_batch.wo)
            recs["wo"] = np.ctypeslib.as_array(C.cast(b.wo, C.POINTER(C.c_uint8)), shape=(int(b.n_aln) * WO_DTYPE.itemsize,)).copy().view(WO_DTYPE)
            recs["wo_runs"] = _runs_of(b)
        return names, descs, off, bases, recs, counts
    finally:
        if g:
            L.pp_ingest_free(g)
        L.pp_assembly_free(a)


_SEQ4_LUT = np.full(256, 15, dtype=np.uint8)
for _c, _v in ((ord("A"), 0), (ord("C"), 1), (ord("T"), 2), (ord("G"), 3), (ord("N"), 4), (ord("-"), 5)):
    _SEQ4_LUT[_c] = _v


def pack_seq4(seq):
    """The 4-bit mirror of a seq array (pp_aln_batch.seq4): base i in bits 4*(i&1).. of byte i >> 1, codes PP_SEQ4_*;
    32 bytes of slack behind, as the header asks for."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    codes = _SEQ4_LUT[seq]
    if codes.size & 1:
        codes = np.concatenate([codes, np.zeros(1, np.uint8)])
    out = np.zeros(codes.size // 2 + 32, dtype=np.uint8)
    out[:codes.size // 2] = codes[0::2] | (codes[1::2] << 4)
    return out


def ingest_device(ctx, assembly, sams, max_errors=10, careful=False, seq_layout=None):
    """The device tokenizer (pp_dev_ingest_*): same return value as ingest(), the records copied back from HBM.
    seq_layout: None = the library's default (window-grouped unless PP_SEQ_LAYOUT=file), 0 = SEQ bytes in file order,
    1 = window-grouped (pp_dev_ingest_set_seq_layout)."""
    L = lib()
    err = C.create_string_buffer(1024)
    a = C.c_void_p()
    rc = L.pp_assembly_load(str(assembly).encode(), C.byref(a), err, 1024)
    if rc:
        raise PolypolishError(rc, err.value.decode())
    g = C.c_void_p()
    try:
        n = L.pp_assembly_n_contigs(a)
        names = [L.pp_assembly_name(a, i).decode() for i in range(n)]
        descs = [L.pp_assembly_description(a, i).decode() for i in range(n)]
        off = np.ctypeslib.as_array(L.pp_assembly_offsets(a), shape=(n + 1,)).copy()
        bases = np.ctypeslib.as_array(L.pp_assembly_bases(a), shape=(int(off[-1]),)).copy()
        ctx._chk(L.pp_dev_ingest_create(ctx._h, a, max_errors, int(careful), C.byref(g)))
        if seq_layout is not None:
            ctx._chk(L.pp_dev_ingest_set_seq_layout(g, int(seq_layout)))
        counts = []
        for s in sams:
            c = SamCounts()
            ctx._chk(L.pp_dev_ingest_sam(g, str(s).encode(), C.byref(c)))
            counts.append((c.alignments, c.used, c.reads))
        b = AlnBatch()
        L.pp_dev_ingest_batch(g, C.byref(b))
        sizes = {"seq": b.seq_bytes, "cigar": b.n_cig_total}
        recs = {}
        for name, dt in REC_FIELDS:
            cnt = int(sizes.get(name, b.n_aln))
            ptr = getattr(b, name)
            arr = np.zeros(cnt, dtype=dt)
            if cnt and ptr:
                ctx._chk(L.pp_ctx_download(ctx._h, arr.ctypes.data, ptr, arr.nbytes))
            recs[name] = arr
        if b.seq4:  # the 4-bit mirror of the seq array (two bases per byte), as the tokenizer hands it to the polish
            m = np.zeros((int(b.seq_bytes) + 1) // 2, dtype=np.uint8)
            if m.size:
                ctx._chk(L.pp_ctx_download(ctx._h, m.ctypes.data, b.seq4, m.nbytes))
            recs["seq4"] = m
        if b.wo and b.n_aln:
            w = np.zeros(int(b.n_aln), dtype=WO_DTYPE)
            ctx._chk(L.pp_ctx_download(ctx._h, w.ctypes.data, b.wo, w.nbytes))
            recs["wo"] = w
            recs["wo_runs"] = _runs_of(b)
        return names, descs, off, bases, recs, counts
    finally:
        if g:
            L.pp_dev_ingest_free(g)
        L.pp_assembly_free(a)


class Plan:
    """pp_shard_plan: which rank polishes which contig / window of a contig (no GPU needed).
    aln_per_contig: alignment records per contig (e.g. np.bincount(recs["contig"], minlength=n_contigs))."""

    def __init__(self, contig_off, aln_per_contig, world, min_window=0):
        self.contig_off = np.ascontiguousarray(contig_off, dtype=np.uint64)
        self.n_contigs = len(self.contig_off) - 1
        cnt = np.ascontiguousarray(aln_per_contig, dtype=np.uint64)
        assert len(cnt) == self.n_contigs
        self._p = C.POINTER(ShardPlan)()
        rc = lib().pp_shard_plan_create(self.n_contigs, self.contig_off.ctypes.data, cnt.ctypes.data, world, min_window,
                                        C.byref(self._p))
        if rc:
            raise PolypolishError(rc, "pp_shard_plan_create failed")
        p = self._p.contents
        n = p.n_units
        self.world = world
        self.unit_contig = np.ctypeslib.as_array(p.contig, shape=(n,)).copy()
        self.unit_lo = np.ctypeslib.as_array(p.lo, shape=(n,)).copy()
        self.unit_hi = np.ctypeslib.as_array(p.hi, shape=(n,)).copy()
        self.unit_rank = np.ctypeslib.as_array(p.rank, shape=(n,)).copy()

    def emit_ranges(self, rank):
        """(n_contigs, 2) array of [lo, hi) the rank emits per contig (pp_polish_set_emit's form)."""
        lo = np.zeros(self.n_contigs, dtype=np.uint64)
        hi = np.zeros(self.n_contigs, dtype=np.uint64)
        rc = lib().pp_shard_emit_ranges(self._p, rank, lo.ctypes.data, hi.ctypes.data)
        if rc:
            raise PolypolishError(rc, "pp_shard_emit_ranges failed")
        return np.stack([lo, hi], axis=1)

    def assemble(self, rank_bytes, rank_contig_off):
        """Polished bytes of all ranks (what each rank's Context.result() gave) -> (bytes in assembly order, offsets)."""
        bufs = [np.frombuffer(b, dtype=np.uint8) if len(b) else np.zeros(1, np.uint8) for b in rank_bytes]
        offs = [np.ascontiguousarray(o, dtype=np.uint64) for o in rank_contig_off]
        bp = (C.c_void_p * self.world)(*[b.ctypes.data for b in bufs])
        op = (C.c_void_p * self.world)(*[o.ctypes.data for o in offs])
        out_off = np.zeros(self.n_contigs + 1, dtype=np.uint64)
        L = lib()
        rc = L.pp_shard_assemble(self._p, bp, op, None, out_off.ctypes.data)
        out = np.zeros(max(int(out_off[-1]), 1), dtype=np.uint8)
        if rc == 0:
            rc = L.pp_shard_assemble(self._p, bp, op, out.ctypes.data, out_off.ctypes.data)
        if rc:
            raise PolypolishError(rc, "pp_shard_assemble failed")
        return out[:int(out_off[-1])].tobytes(), out_off

    def close(self):
        if self._p:
            lib().pp_shard_plan_free(self._p)
            self._p = C.POINTER(ShardPlan)()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardPart:
    """pp_shard_split: the records of a batch that rank `dest` of a plan needs (its contigs' records; on a tiled contig
    its window's records plus those that reach in from the neighbours).  The batch is given as raw pointers
    (`ptrs`: field name -> address, host or device as `mem` says); `ctx` may be None for host memory.
      .n_aln, .seq_bytes, .n_cig_total, .ptrs (field name -> address of the part's arrays, same memory kind), .orig_ptr
      .host() -> (recs dict of numpy arrays, orig) for a host part."""

    def __init__(self, ctx, plan, dest, n_aln, ptrs, seq_bytes, n_cig_total, mem):
        L = lib()
        b = aln_batch(n_aln, ptrs, seq_bytes, n_cig_total)  # (ptrs["wo_runs"]: the mirror's runs go along, restricted)
        self._p = C.c_void_p()
        self._ctx = ctx  # keeps the context (and with it the part's device memory) alive
        rc = L.pp_shard_split(ctx._h if ctx is not None else None, plan._p, dest, C.byref(b), mem, C.byref(self._p))
        if rc:
            raise PolypolishError(rc, L.pp_last_error(ctx._h).decode() if ctx is not None else "pp_shard_split failed")
        out, orig = AlnBatch(), C.c_void_p()
        L.pp_shard_part_batch(self._p, C.byref(out), C.byref(orig))
        self.mem = mem
        self.n_aln, self.seq_bytes, self.n_cig_total = int(out.n_aln), int(out.seq_bytes), int(out.n_cig_total)
        self.ptrs = {name: (C.cast(getattr(out, name), C.c_void_p).value or 0) for name, _ in REC_FIELDS}
        if out.seq4:  # a device part brings the 4-bit mirror of its seq array
            self.ptrs["seq4"] = out.seq4
        if out.wo:    # ... and every part the window-order mirror of its records, when the source batch has one
            self.ptrs["wo"] = out.wo
            runs = _runs_of(out)
            if len(runs):  # ... with its runs, when the source's are known: the part takes the direct path too
                self.ptrs["wo_runs"] = runs
        self.orig_ptr = orig.value or 0

    def host(self):
        assert self.mem == MEM_HOST
        sizes = {"seq": self.seq_bytes, "cigar": self.n_cig_total}
        recs = {}
        for name, dt in REC_FIELDS:
            cnt = int(sizes.get(name, self.n_aln))
            recs[name] = (np.ctypeslib.as_array(C.cast(self.ptrs[name], C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()
                          if cnt and self.ptrs[name] else np.zeros(0, dtype=dt))
        orig = (np.ctypeslib.as_array(C.cast(self.orig_ptr, C.POINTER(C.c_uint32)), shape=(self.n_aln,)).copy()
                if self.n_aln else np.zeros(0, np.uint32))
        if self.ptrs.get("wo") and self.n_aln:
            recs["wo"] = np.ctypeslib.as_array(C.cast(self.ptrs["wo"], C.POINTER(C.c_uint8)), shape=(self.n_aln * WO_DTYPE.itemsize,)).copy().view(WO_DTYPE)
        return recs, orig

    def close(self):
        if self._p:
            lib().pp_shard_part_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_split_host(plan, dest, recs):
    """Host numpy SoA -> (recs of the part, orig).  recs["wo"] (the window-order mirror, WO_DTYPE) goes along when it is there,
    and recs["wo_runs"] (the ends of its runs) with it."""
    keep = {k: np.ascontiguousarray(recs[k], dtype=dt) for k, dt in REC_FIELDS}
    ptrs = {k: v.ctypes.data for k, v in keep.items()}
    if "wo" in recs and len(recs["wo"]):
        keep["wo"] = np.ascontiguousarray(recs["wo"])
        ptrs["wo"] = keep["wo"].ctypes.data
        if recs.get("wo_runs") is not None and len(recs["wo_runs"]):  # the mirror's runs: restricted along with it
            ptrs["wo_runs"] = recs["wo_runs"]
    part = ShardPart(None, plan, dest, len(keep["contig"]), ptrs, len(keep["seq"]),
                     len(keep["cigar"]), MEM_HOST)
    out = part.host()
    if "wo_runs" in part.ptrs and "wo" in out[0]:
        out[0]["wo_runs"] = part.ptrs["wo_runs"]
    part.close()
    return out


def shard_count(ctx, n_aln, contig_ptr, mem, n_contigs, ptrs=None):
    """pp_shard_count: alignment records per contig of a batch given by pointers."""
    p = ptrs or {}
    one = contig_ptr
    b = AlnBatch(n_aln, contig_ptr, p.get("ref_start", one), p.get("k", one), p.get("seq_off", one), p.get("seq_len", one),
                 p.get("cig_off", one), p.get("n_cig", one), p.get("seq", one), 0, p.get("cigar", one), 0)
    cnt = np.zeros(n_contigs, dtype=np.uint64)
    rc = lib().pp_shard_count(ctx._h if ctx is not None else None, C.byref(b), mem, n_contigs, cnt.ctypes.data)
    if rc:
        raise PolypolishError(rc, "pp_shard_count failed")
    return cnt


def comm_unique_id() -> bytes:
    """ncclUniqueId for pp_comm_init (rank 0 makes it, the launcher hands it to every rank)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().pp_comm_unique_id(buf)
    if rc:
        raise PolypolishError(rc, "pp_comm_unique_id failed (librccl not loadable?)")
    return buf.raw


class FilterLoaded:
    """Host half of the filter (pp_filter_load / pp_filter_write; no GPU needed): `files[f]` holds the SoA
    of pp_filter_file as numpy copies, `counts[f]` = (alignments, distinct read names)."""

    def __init__(self, in1, in2):
        L = lib()
        self._h = C.c_void_p()
        err = C.create_string_buffer(1400)
        fc = (FilterFileCounts * 2)()
        rc = L.pp_filter_load(str(in1).encode(), str(in2).encode(), C.byref(self._h), fc, err, 1400)
        self.counts = [(c.alignments, c.reads) if c.loaded else None for c in fc]
        if rc:
            self._h = C.c_void_p()
            raise PolypolishError(rc, err.value.decode())
        self.input = FilterInput()
        L.pp_filter_loaded_input(self._h, C.byref(self.input))
        self.n_reads = int(self.input.n_reads)
        self.files = []
        for f in range(2):
            d = self.input.file[f]
            n = int(d.n_aln)

            def arr(ptr, cnt, dt):
                if not cnt or not ptr:
                    return np.zeros(0, dtype=dt)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(cnt,)).copy()
            self.files.append({
                "ref_id": arr(d.ref_id, n, np.uint32), "ref_start": arr(d.ref_start, n, np.uint32),
                "flags": arr(d.flags, n, np.uint32), "cig_off": arr(d.cig_off, n, np.uint64),
                "n_cig": arr(d.n_cig, n, np.uint32), "cigar": arr(d.cigar, int(d.n_cig_total), np.uint32),
                "read": arr(d.read, n, np.uint32), "grp_off": arr(d.grp_off, self.n_reads + 1, np.uint32),
                "grp_idx": arr(d.grp_idx, n, np.uint32)})

    def write(self, f, passed, path):
        passed = np.ascontiguousarray(passed, dtype=np.uint8)
        err = C.create_string_buffer(1400)
        p_, f_ = C.c_uint64(), C.c_uint64()
        rc = lib().pp_filter_write(self._h, f, passed.ctypes.data, str(path).encode(), C.byref(p_), C.byref(f_), err, 1400)
        if rc:
            raise PolypolishError(rc, err.value.decode())
        return p_.value, f_.value

    def close(self):
        if self._h:
            lib().pp_filter_loaded_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FilterLoadedDevice:
    """pp_filter_load_device: the same load on the GPU; `files[f]` holds copies of the device arrays
    (ref_end instead of the CIGAR arrays), `counts[f]` = (alignments, distinct read names)."""

    def __init__(self, ctx, in1, in2):
        L = lib()
        self._h = C.c_void_p()
        fc = (FilterFileCounts * 2)()
        rc = L.pp_filter_load_device(ctx._h, str(in1).encode(), str(in2).encode(), C.byref(self._h), fc)
        self.counts = [(c.alignments, c.reads) if c.loaded else None for c in fc]
        if rc:
            self._h = C.c_void_p()
            raise PolypolishError(rc, L.pp_last_error(ctx._h).decode())
        self.input = FilterInput()
        L.pp_filter_dev_input(self._h, C.byref(self.input))
        self.n_reads = int(self.input.n_reads)
        self.files = []
        for f in range(2):
            d = self.input.file[f]
            n = int(d.n_aln)

            def arr(ptr, cnt, dt):
                a = np.zeros(cnt, dtype=dt)
                if cnt and ptr:
                    ctx._chk(L.pp_ctx_download(ctx._h, a.ctypes.data, ptr, a.nbytes))
                return a
            self.files.append({
                "ref_id": arr(d.ref_id, n, np.uint32), "ref_start": arr(d.ref_start, n, np.uint32),
                "flags": arr(d.flags, n, np.uint32), "ref_end": arr(d.ref_end, n, np.uint64),
                "read": arr(d.read, n, np.uint32), "grp_off": arr(d.grp_off, self.n_reads + 1, np.uint32),
                "grp_idx": arr(d.grp_idx, n, np.uint32)})

    def close(self):
        if self._h:
            lib().pp_filter_dev_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One HIP device + stream (pp_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().pp_ctx_create(device, C.byref(self._h))
        if rc:
            raise PolypolishError(rc, f"pp_ctx_create({device}) failed: no usable HIP device (there is no CPU path)")
        self._keep = []

    def close(self):
        if self._h:
            lib().pp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise PolypolishError(rc, lib().pp_last_error(self._h).decode())

    def sync(self):
        self._chk(lib().pp_ctx_sync(self._h))

    def set_profiling(self, on=True):
        """False/0 off, True/1 every kernel group, 2 only the dominant kernel (one event pair per job)."""
        lib().pp_ctx_set_profiling(self._h, int(on))

    # ---- seam B -------------------------------------------------------------------------------
    def polish_begin(self, contig_off, bases_ptr, bases_mem, min_depth=5, fraction_valid=0.5, fraction_invalid=0.2):
        off = np.ascontiguousarray(contig_off, dtype=np.uint64)
        p = Params(min_depth, fraction_valid, fraction_invalid)
        self._n_contigs = len(off) - 1
        self._G = int(off[-1])
        self._chk(lib().pp_polish_begin(self._h, self._n_contigs, off.ctypes.data, bases_ptr, bases_mem, C.byref(p)))

    def polish_add_ptrs(self, n_aln, ptrs: dict, seq_bytes, n_cig_total, mem):
        b = aln_batch(n_aln, ptrs, seq_bytes, n_cig_total)
        self._chk(lib().pp_polish_add(self._h, C.byref(b), mem))

    def polish_finish(self):
        self._chk(lib().pp_polish_finish(self._h))

    def prepared_job(self, contig_off, bases_ptr, bases_mem, n_aln, ptrs: dict, seq_bytes, n_cig_total, mem,
                     min_depth=5, fraction_valid=0.5, fraction_invalid=0.2, emit=None):
        """begin + (set_emit) + add + finish of ONE job with every argument marshalled once: returns run(), which makes
        the three (four) C calls and nothing else.  For callers that repeat a job on resident data (bench.py's steps): the
        Python side of polish_begin / polish_add_ptrs costs tens of microseconds per call, during which the GPU idles."""
        L = lib()
        off = np.ascontiguousarray(contig_off, dtype=np.uint64)
        p = Params(min_depth, fraction_valid, fraction_invalid)
        b = aln_batch(n_aln, ptrs, seq_bytes, n_cig_total)
        n_contigs, G = len(off) - 1, int(off[-1])
        h, off_p, p_ref, b_ref = self._h, off.ctypes.data, C.byref(p), C.byref(b)
        begin, add, finish, set_emit = L.pp_polish_begin, L.pp_polish_add, L.pp_polish_finish, L.pp_polish_set_emit
        if emit is not None:
            e = np.ascontiguousarray(emit, dtype=np.uint64).reshape(n_contigs, 2)
            lo, hi = np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])
            lo_p, hi_p = lo.ctypes.data, hi.ctypes.data
        keep = (off, p, b, emit is not None and (lo, hi))  # the C side reads these during the calls

        def run():
            self._n_contigs, self._G = n_contigs, G
            rc = begin(h, n_contigs, off_p, bases_ptr, bases_mem, p_ref)
            if rc == 0 and emit is not None:
                rc = set_emit(h, lo_p, hi_p)
            if rc == 0:
                rc = add(h, b_ref, mem)
            if rc == 0:
                rc = finish(h)
            if rc:
                self._chk(rc)
        run._keep = keep
        return run

    def result_size(self):
        n = C.c_uint64()
        self._chk(lib().pp_polish_result_size(self._h, C.byref(n)))
        return n.value

    def result_device_ptr(self):
        return lib().pp_polish_result_device(self._h)

    def result(self):
        n = self.result_size()
        out = np.zeros(max(n, 1), dtype=np.uint8)
        offs = np.zeros(self._n_contigs + 1, dtype=np.uint64)
        stats = (ContigStats * self._n_contigs)()
        self._chk(lib().pp_polish_result(self._h, out.ctypes.data, MEM_HOST, offs.ctypes.data, stats))
        st = [dict(polished_len=s.polished_len, changed=s.changed, zero_depth=s.zero_depth, depth_sum=s.depth_sum)
              for s in stats]
        return out[:n].tobytes(), offs, st

    def trust_mirrors(self, on=True):
        """(internal hook, pp_ctx_trust_mirrors_) every window-order mirror this context is given counts as one of the library's
        own: it is not compared with the arrays before the kernels read the records through it.  For callers that lay their
        batches out exactly as the library's ingests do (bench.py's resident job; tests/test_synthjob_cpu.py checks that it does)."""
        f = lib().pp_ctx_trust_mirrors_
        f.argtypes = [C.c_void_p, C.c_int]
        f.restype = C.c_int
        self._chk(f(self._h, int(bool(on))))

    def took_direct_path(self):
        """True when the last pp_polish_finish took its bulk straight from the window-order mirror (pp_aln_batch.wo_run_end)."""
        return bool(lib().pp_polish_took_direct_path(self._h))

    def kernel_times(self):
        kt = KernelTimes()
        lib().pp_polish_kernel_times(self._h, C.byref(kt))
        return kt.as_dict()

    def first_kernel_ms(self):
        """(name as bytes, milliseconds) of the first kernel group the last job timed, or None: what bench.py's timed steps read
        after every job (profiling level 2: the dominant kernel alone) -- one C call into a structure that is kept, no dictionaries."""
        kt = getattr(self, "_kt", None)
        if kt is None:
            kt = self._kt = KernelTimes()
            self._kt_ref = C.byref(kt)
            self._kt_fn = lib().pp_polish_kernel_times
        self._kt_fn(self._h, self._kt_ref)
        return (kt.name[0], kt.ms[0]) if kt.n else None

    def positions(self):
        G = self._G
        arrs = {"depth": np.zeros(G, np.float64), "status": np.zeros(G, np.uint8)}
        for k in ("count_a", "count_c", "count_g", "count_t", "count_other", "valid_thr", "invalid_thr"):
            arrs[k] = np.zeros(G, np.uint32)
        po = PositionsOut(*[arrs[k].ctypes.data for k in ("depth", "count_a", "count_c", "count_g", "count_t",
                                                          "count_other", "valid_thr", "invalid_thr", "status")])
        self._chk(lib().pp_polish_positions(self._h, C.byref(po)))
        return arrs

    def set_emit(self, emit):
        """emit: None or an (n_contigs, 2) array of [lo, hi) positions each contig emits (pp_polish_set_emit)."""
        if emit is None:
            self._chk(lib().pp_polish_set_emit(self._h, None, None))
            return
        e = np.ascontiguousarray(emit, dtype=np.uint64).reshape(-1, 2)
        lo, hi = np.ascontiguousarray(e[:, 0]), np.ascontiguousarray(e[:, 1])
        if len(lo) != self._n_contigs:
            raise PolypolishError(ERR_ARG, "set_emit: one range per contig")
        self._chk(lib().pp_polish_set_emit(self._h, lo.ctypes.data, hi.ctypes.data))

    def comm_init(self, rank, world, unique_id: bytes):
        """Join the RCCL communicator of the job (pp_comm_init): one context per rank / GPU."""
        self._chk(lib().pp_comm_init(self._h, rank, world, C.c_char_p(unique_id)))
        self._comm = (rank, world)

    def gather(self, gathered_ptr, cap):
        """pp_polish_gather: the polished bytes of all ranks -> rank 0's DEVICE buffer (rank-major).
        Returns (rank_len, rank_contig_off) as numpy arrays (on every rank)."""
        rank, world = self._comm
        lens = np.zeros(world, dtype=np.uint64)
        offs = np.zeros((world, self._n_contigs + 1), dtype=np.uint64)
        self._chk(lib().pp_polish_gather(self._h, gathered_ptr, cap, lens.ctypes.data, offs.ctypes.data))
        return lens, offs

    def polish_records(self, contig_off, bases, recs, min_depth=5, fraction_valid=0.5, fraction_invalid=0.2,
                       positions=False, emit=None, cuts=None):
        """Host numpy SoA (field names of pp_aln_batch) -> polished bytes, offsets, stats.
        cuts: optional record indices at which the records are cut into several pp_polish_add batches.
        positions: True = the per-position records of --debug; 3 = the records of what the pileup kernel itself decides
        (test hook: positions with inexact depth shares settled by its interval test are not replayed)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        keep = {k: np.ascontiguousarray(recs[k], dtype=dt) for k, dt in REC_FIELDS}
        lib().pp_polish_set_debug(self._h, int(positions))
        self.polish_begin(contig_off, bases.ctypes.data, MEM_HOST, min_depth, fraction_valid, fraction_invalid)
        if emit is not None:
            self.set_emit(emit)
        for part in split_records(keep, cuts):
            ptrs = {k: v.ctypes.data for k, v in part.items()}
            if cuts is None and recs.get("wo") is not None and len(recs["wo"]) == len(keep["contig"]) and len(recs["wo"]):
                # the window-order mirror of the records (WO_DTYPE) and, optionally, the ends of its runs
                keep["wo"] = np.ascontiguousarray(recs["wo"])
                ptrs["wo"] = keep["wo"].ctypes.data
                if recs.get("wo_runs") is not None and len(recs["wo_runs"]):
                    ptrs["wo_runs"] = recs["wo_runs"]
            self.polish_add_ptrs(len(part["contig"]), ptrs, len(part["seq"]), len(part["cigar"]), MEM_HOST)
        self.polish_finish()
        polished, offs, stats = self.result()
        res = {"polished": polished, "offsets": offs, "stats": stats, "positions": None}
        if positions:
            res["positions"] = self.positions()
        lib().pp_polish_set_debug(self._h, 0)
        return res

    # ---- whole commands -----------------------------------------------------------------------
    def polish_files(self, assembly, sams, fraction_invalid=0.2, fraction_valid=0.5, max_errors=10, min_depth=5,
                     careful=False, debug=None, quiet=True):
        opt = PolishOptions(fraction_invalid, fraction_valid, max_errors, min_depth, int(careful),
                            str(debug).encode() if debug else None, int(quiet))
        arr = (C.c_char_p * max(len(sams), 1))(*[str(s).encode() for s in sams])
        out = Bytes()
        self._chk(lib().pp_polish_files(self._h, str(assembly).encode(), arr, len(sams), C.byref(opt), C.byref(out)))
        data = C.string_at(out.data, out.len) if out.len else b""
        lib().pp_bytes_free(C.byref(out))
        return data

    def filter_polish_files(self, assembly, in1, in2, out1=None, out2=None, orientation="auto", low=0.1, high=99.9,
                            fraction_invalid=0.2, fraction_valid=0.5, max_errors=10, min_depth=5, careful=False,
                            debug=None, quiet=True):
        """filter + polish in one process (pp_filter_polish_files): returns (FASTA bytes, filter report)."""
        opt = PolishOptions(fraction_invalid, fraction_valid, max_errors, min_depth, int(careful),
                            str(debug).encode() if debug else None, int(quiet))
        rep, out = FilterReport(), Bytes()
        enc = lambda x: str(x).encode() if x is not None else None
        self._chk(lib().pp_filter_polish_files(self._h, enc(assembly), enc(in1), enc(in2), enc(out1), enc(out2),
                                               orientation.encode(), low, high, C.byref(opt), C.byref(rep), C.byref(out)))
        data = C.string_at(out.data, out.len) if out.len else b""
        lib().pp_bytes_free(C.byref(out))
        return data, {"before": rep.before_count, "after": rep.after_count, "low": rep.low_threshold,
                      "high": rep.high_threshold, "orientation": ("fr", "rf", "ff", "rr")[rep.orientation],
                      "counts": list(rep.orientation_counts)}

    def filter_files(self, in1, in2, out1, out2, orientation="auto", low=0.1, high=99.9, quiet=True):
        rep = FilterReport()
        self._chk(lib().pp_filter_files(self._h, str(in1).encode(), str(in2).encode(), str(out1).encode(),
                                        str(out2).encode(), orientation.encode(), low, high, int(quiet),
                                        C.byref(rep)))
        return {"before": rep.before_count, "after": rep.after_count, "low": rep.low_threshold,
                "high": rep.high_threshold, "orientation": ("fr", "rf", "ff", "rr")[rep.orientation],
                "counts": list(rep.orientation_counts)}


_default_ctx = None


def _ctx():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("PP_DEVICE", "0")))
    return _default_ctx


def polish(assembly, sam, debug=None, fraction_invalid=0.2, fraction_valid=0.5, max_errors=10, min_depth=5,
           careful=False) -> bytes:
    """polish::polish (src/polish.rs:26-38): returns the FASTA bytes the reference prints to stdout."""
    return _ctx().polish_files(assembly, list(sam), fraction_invalid, fraction_valid, max_errors, min_depth, careful,
                               debug)


def filter(in1, in2, out1, out2, orientation="auto", low=0.1, high=99.9):  # noqa: A001 (reference name)
    """filter::filter (src/filter.rs:26-37): writes out1/out2, returns the before/after report."""
    return _ctx().filter_files(in1, in2, out1, out2, orientation, low, high)
