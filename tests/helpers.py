"""Shared comparison helpers for the parity tests."""
import numpy as np

from oracle import pyoracle as po
from uncalled_amd import capi

HIT_INT_FIELDS = ("mapped", "fwd", "rd_st", "rd_en", "rd_len", "rf_st", "rf_en", "rf_len", "matches",
                  "n_events", "event_i", "n_nbr", "n_sa", "n_lf", "notes")


def oracle_hits(oix, raw, offsets, calib, params=None, fresh_mapper_per_read=False):
    """Map every read with the oracle restatement; returns O_HIT array."""
    out = np.zeros(offsets.size - 1, dtype=po.O_HIT)
    om = po.Mapper(oix, params)
    for i in range(offsets.size - 1):
        if fresh_mapper_per_read:
            om = po.Mapper(oix, params)
        sig = po.calibrate(raw[int(offsets[i]):int(offsets[i + 1])], float(calib["range"][i]),
                           float(calib["offset"][i]), float(calib["digitisation"][i]))
        out[i] = om.map_read(sig)
    return out


def assert_hits_equal(dev_hits, ora_hits, what=""):
    """Bit-exact: PAF coordinates, event counts and SURVEY 8(d) work counters; mean_event_len within 0 ulp."""
    assert len(dev_hits) == len(ora_hits)
    for i, (d, o) in enumerate(zip(dev_hits, ora_hits)):
        assert int(d["status"]) == 0, f"{what} read {i}: device status {int(d['status'])}"
        for f in HIT_INT_FIELDS:
            assert int(d[f]) == int(o[f]), f"{what} read {i}: {f} device={int(d[f])} oracle={int(o[f])}"
        if d["mapped"]:
            assert int(d["rid"]) == int(o["rid"]), f"{what} read {i}: rid"
            assert (int(d["cl_ref_st"]), int(d["cl_ref_en_start"]), int(d["cl_ref_en_end"]), int(d["cl_evt_st"]),
                    int(d["cl_evt_en"]), int(d["cl_total_len"])) == \
                   (int(o["cluster"]["ref_st"]), int(o["cluster"]["ref_en_start"]), int(o["cluster"]["ref_en_end"]),
                    int(o["cluster"]["evt_st"]), int(o["cluster"]["evt_en"]), int(o["cluster"]["total_len"])), \
                   f"{what} read {i}: cluster"
        assert np.float32(d["mean_event_len"]) == np.float32(o["mean_event_len"]) or int(o["n_events"]) == 0, \
            f"{what} read {i}: mean_event_len"


def to_oracle_params(p):
    """capi.Params -> pyoracle.Params (same field names)."""
    q = po.default_params()
    for name, _ in q._fields_:
        setattr(q, name, getattr(p, name))
    return q


def oracle_hits_threads(oix, raw, offsets, calib, dev_hits=None, threads=None):
    """The oracle on all host threads (po.map_batch: every thread ONE Mapper that maps read after read) for batches of thousands
    of reads.  A thread's Mapper carries sources_added_ from a read that filled max_paths into its next read (mapper.cpp:88,612-623),
    whichever read that happens to be; the device maps every read as a fresh Mapper does.  So a read that disagrees with `dev_hits`
    is mapped once more by a FRESH oracle Mapper before it counts (tests/dev/parity_sweep.py does the same against the reference)."""
    import os
    threads = threads or max(1, min(offsets.size - 1, len(os.sched_getaffinity(0))))
    sig = po.calibrate(raw, float(calib["range"][0]), float(calib["offset"][0]), float(calib["digitisation"][0]))
    want, secs = po.map_batch(oix, sig, offsets, threads)
    redone = 0
    if dev_hits is not None:
        for i in range(offsets.size - 1):
            if any(int(dev_hits[f][i]) != int(want[f][i]) for f in HIT_INT_FIELDS):
                want[i] = po.Mapper(oix).map_read(sig[int(offsets[i]):int(offsets[i + 1])])
                redone += 1
    return want, secs, redone
