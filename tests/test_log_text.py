"""The three strings of the run log that the reference pins with unit tests (get_percentile_name, format_duration,
qscore), checked on the PRODUCT's own host code through pp_log_text -- CPU only, the library loads without a GPU."""
import ctypes as C
import json
import os

import polypolish_amd as pp

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_unit_vectors.json"), encoding="utf-8"))
QSCORE, DURATION, PERCENTILE_NAME = 0, 1, 2


def text(what, value):
    buf = C.create_string_buffer(128)
    assert pp.lib().pp_log_text(what, float(value), buf, len(buf)) == 0
    return buf.value.decode("utf-8")


def test_T7_percentile_name():
    for p, want in GOLD["T7_percentile_name"]["cases"]:
        assert text(PERCENTILE_NAME, p) == want


def test_percentile_name_follows_f64_to_string():
    # Rust formats the f64 with its shortest round-trip decimal form, never with an exponent
    assert text(PERCENTILE_NAME, 0.123456789) == "0.123456789th percentile"
    assert text(PERCENTILE_NAME, 0.00001) == "0.00001st percentile"
    assert text(PERCENTILE_NAME, 11.0) == "11th percentile"
    assert text(PERCENTILE_NAME, 12.0) == "12th percentile"
    assert text(PERCENTILE_NAME, 13.0) == "13th percentile"
    assert text(PERCENTILE_NAME, 21.0) == "21st percentile"
    assert text(PERCENTILE_NAME, 99.25) == "99.25th percentile"


def test_T12_format_duration():
    for us, want in GOLD["T12_format_duration"]["cases"]:
        assert text(DURATION, us) == want


def test_T13_qscore():
    for identity, want in GOLD["T13_qscore"]["cases"]:
        assert text(QSCORE, identity) == want


def test_bad_arguments():
    buf = C.create_string_buffer(4)
    assert pp.lib().pp_log_text(PERCENTILE_NAME, 99.9, buf, len(buf)) != 0  # does not fit
    assert pp.lib().pp_log_text(7, 1.0, buf, len(buf)) != 0
