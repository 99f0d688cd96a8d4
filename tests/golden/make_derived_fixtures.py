#!/usr/bin/env python3
"""Regenerates tests/golden/derived_e2e/: a small paired SAM + assembly (tests/synth.py, fixed seed) and what the
oracle makes of them -- `filter` (tagged SAMs) then `polish` (FASTA), plus `polish` of the unfiltered SAMs.
DERIVED fixtures: they pin the oracle (and through it the product) against drift; they are not reference-supplied
(the reference cannot be built in this image).  Usage: python tests/golden/make_derived_fixtures.py"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from oracle import orc  # noqa: E402

out = os.path.join(HERE, "derived_e2e")
shutil.rmtree(out, ignore_errors=True)
os.makedirs(out)
ds = synth.rich_dataset(out, seed=4242, contig_lens=(1800, 700), coverage=14, repeat_len=250, repeat_copies=3, zp_frac=0.03,
                        lowercase_frac=0.1, prefix="case")
f1, f2 = os.path.join(out, "filtered_1.sam"), os.path.join(out, "filtered_2.sam")
rep = orc.filter_files(ds["sam1"], ds["sam2"], f1, f2)
polished_filtered = orc.polish_files(ds["fasta"], [f1, f2])
polished_raw = orc.polish_files(ds["fasta"], [ds["sam1"], ds["sam2"]], careful=True, min_depth=3)
open(os.path.join(out, "polished_after_filter.fasta"), "wb").write(polished_filtered["fasta"])
open(os.path.join(out, "polished_raw_careful_d3.fasta"), "wb").write(polished_raw["fasta"])
sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
json.dump({"filter_report": rep, "filtered_1_sha256": sha(f1), "filtered_2_sha256": sha(f2),
           "counts_after_filter": list(polished_filtered["counts"]), "counts_raw": list(polished_raw["counts"])},
          open(os.path.join(out, "expected.json"), "w"), indent=1)
os.remove(f1)
os.remove(f2)
print("wrote", sorted(os.listdir(out)))
