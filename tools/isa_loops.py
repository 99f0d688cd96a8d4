#!/usr/bin/env python3
"""The loops of one kernel in the compiler's assembly listing, with their static instruction mix -- what round 5's look at
k_tile_direct's item loop was made from.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ipolypolish_amd/csrc -mllvm -disable-machine-licm \\
          -S --cuda-device-only polypolish_amd/csrc/pp_kernels.hip -o /tmp/pp_kernels.s
    python tools/isa_loops.py /tmp/pp_kernels.s _ZN2pp13k_tile_directENS_8TileArgsE [min_lines]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
min_lines = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lo = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
hi = next(i for i in range(lo, len(lines)) if "s_endpgm" in lines[i])
labels = {m.group(1): i for i in range(lo, hi) for m in [re.match(r"^(\.LBB[0-9_]+):", lines[i])] if m}
loops = set()
for i in range(lo, hi):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB[0-9_]+)", lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.add((labels[m.group(1)], i))
def mix(a, b):
    c = lambda pat: sum(1 for l in lines[a:b] if re.match(pat, l))
    return dict(valu=c(r"\s+v_"), salu=c(r"\s+s_(?!waitcnt|nop|cbranch|branch|barrier)"), vmem_ld=c(r"\s+(global|flat|buffer)_load"),
                vmem_st=c(r"\s+(global|flat|buffer)_(store|atomic)"), lds=c(r"\s+ds_"), lds_atomic=c(r"\s+ds_add"), waits=c(r"\s+s_waitcnt"))
tot = mix(lo, hi)
print(f"{sym}: {hi - lo} lines, {tot}")
print("loops (first line, last line of the listing; innermost ones are the short ones):")
for a, b in sorted(loops):
    if b - a >= min_lines:
        print(f"  {a + 1:6d} .. {b + 1:6d}  {b - a:5d} lines  {mix(a, b)}")
