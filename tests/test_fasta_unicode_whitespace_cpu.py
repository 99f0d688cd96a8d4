"""`text[1..].splitn(2, char::is_whitespace)` (src/misc.rs:118-120): Rust's char::is_whitespace is the Unicode White_Space
property -- a FASTA header splits at U+00A0, U+2003, U+3000 ... too, the whole character is dropped, and the line the reference
prints is `>name description polypolish` (src/polish.rs:196-202).  The product's loader and the oracle's, against the splits
the property gives, and against each other through the polished FASTA of a job without alignments.  (the
oracle used to split at ASCII whitespace only.)"""
import ctypes as C


WHITE_SPACE = [0x20, 0x09, 0x0B, 0x0C, 0x85, 0xA0, 0x1680, 0x2000, 0x2003, 0x200A, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000]
NOT_WHITE_SPACE = [0x200B, 0xAD, 0x2060, 0xFEFF, 0x180E]   # zero-width space, soft hyphen, word joiner, BOM, Mongolian vowel separator


def test_fasta_header_splits_at_unicode_whitespace_like_rust(orc, tmp_path):
    import polypolish_amd as pp
    lines, want = [], []
    for i, cp in enumerate(WHITE_SPACE):
        s = chr(cp)
        lines.append(f">c{i}{s}first{s}second\nACGT\n")
        want.append((f"c{i}", f"first{s}second"))
    for i, cp in enumerate(NOT_WHITE_SPACE):
        s = chr(cp)
        lines.append(f">n{i}{s}x y\nACGT\n")
        want.append((f"n{i}{s}x", "y"))
    fa = tmp_path / "u.fasta"
    fa.write_bytes("".join(lines).encode("utf-8"))
    L = pp.lib()
    a, err = C.c_void_p(), C.create_string_buffer(512)
    assert L.pp_assembly_load(str(fa).encode(), C.byref(a), err, 512) == 0, err.value
    try:
        got = [(L.pp_assembly_name(a, i).decode("utf-8"), L.pp_assembly_description(a, i).decode("utf-8"))
               for i in range(L.pp_assembly_n_contigs(a))]
    finally:
        L.pp_assembly_free(a)
    assert got == want
    out = orc.polish_files(str(fa), [])["fasta"].decode("utf-8").split("\n")
    assert out[0::2][:len(want)] == [f">{n} {d} polypolish" for n, d in want]
