"""uncalled_amd.pafstats against reports printed by the reference's uncalled/pafstats.py (tests/golden/pafstats/, made by
tests/golden/make_pafstats_goldens.py)."""
import argparse
import io
from pathlib import Path

import pytest

from uncalled_amd import pafstats

G = Path(__file__).resolve().parent / "golden" / "pafstats"

CASES = {"plain": [], "first20": ["-n", "20"], "ref": ["-r", str(G / "truth.paf")], "annotate": ["-r", str(G / "truth.paf"), "-a"]}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_report_matches_reference(tag):
    ap = argparse.ArgumentParser()
    pafstats.add_opts(ap)
    args = ap.parse_args([str(G / "query.paf")] + CASES[tag])
    out, err = io.StringIO(), io.StringIO()
    pafstats.run(args, out=out, err=err)
    assert out.getvalue() == (G / (tag + ".stdout")).read_text()
    assert err.getvalue() == (G / (tag + ".stderr")).read_text()


def test_entry_roundtrip_and_geometry():
    lines = [l for l in (G / "query.paf").read_text().splitlines() if l]
    for l in lines:
        e = pafstats.PafEntry(l)
        n_same = 12 if e.is_mapped else 2   # unmapped records print ten '*' columns (pafstats.py:103), qual included
        assert str(e).split("\t")[:n_same] == l.split("\t")[:n_same]
        if e.is_mapped:
            r = e.rev()
            assert r.qry_loc() == e.ref_loc() and r.ref_loc() == e.qry_loc()
            lo, hi = e.ext_ref(1.0)
            assert 1 <= lo <= e.rf_st and e.rf_en <= hi <= e.rf_len
            assert e.contains(e) and e.overlaps(e)
        else:
            assert e.qr_st == 1 and e.qr_en == e.qr_len
