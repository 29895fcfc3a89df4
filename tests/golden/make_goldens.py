#!/usr/bin/env python3
"""Generates tests/golden/ref_goldens.npz from oracle/_ref (the reference's own hot-path sources compiled
in place): stage-level and PAF-level known answers for
  * the bundled example read (events, normalised levels, match log-probs of the first events, PAF), and
  * 48 seeded synthetic reads against the bundled example index (signals stored, PAF + work counters).
Container-only: needs /root/reference (via `make -C oracle ref`).  The outputs are committed so that the
oracle restatement and the HIP path can be checked where the reference does not exist (GPU box).
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pyref  # noqa: E402
from uncalled_amd.build_index import encode_contigs, read_fasta  # noqa: E402
from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE, simulate_reads  # noqa: E402

G = Path(__file__).resolve().parent
PREFIX = G / "example_index" / "example_ref"


def hit_tuple(h):
    return np.array([h.mapped, h.fwd, h.rd_st, h.rd_en, h.rd_len, h.rf_st, h.rf_en, h.rf_len, h.matches,
                     h.n_events, h.event_i, h.n_nbr, h.n_sa, h.n_lf], dtype=np.int64)


def main():
    pyref.init(PREFIX)
    ex = np.load(G / "example_read.npz")
    sig = pyref.calibrate(ex["signal"], float(ex["range"]), float(ex["offset"]), float(ex["digitisation"]))
    ev, mel, tot = pyref.events(sig)
    levels, scale, shift = pyref.norm_levels(ev["mean"])
    probs = np.stack([pyref.match_probs(x) for x in levels[:64]])
    m = pyref.Mapper()
    h = m.map_read(sig)
    names, _, seqs = read_fasta(str(PREFIX) + ".fa")
    codes, _, _ = encode_contigs(seqs)
    sim = simulate_reads(codes, [len(s) for s in seqs], 48, seed=2024, read_bases=1500)
    hits, mels = [], []
    for i in range(48):
        raw = sim["signal"][int(sim["offsets"][i]):int(sim["offsets"][i + 1])]
        hh = m.map_read(pyref.calibrate(raw, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION))
        hits.append(hit_tuple(hh))
        mels.append(hh.mean_event_len)
    # chunked path (Mapper::new_read(Chunk)/add_chunk/process_chunk/map_chunk as MapPoolOrd drives it), all reads
    # one after the other on ONE Mapper = one channel, so the rolling normaliser carries over between reads
    cm = pyref.Mapper()
    chunk_hits, chunk_used = [], []
    hh, used = cm.chunk_read(sig, 4000, 0)
    chunk_hits.append(hit_tuple(hh)); chunk_used.append(used)
    for i in range(30):
        raw = sim["signal"][int(sim["offsets"][i]):int(sim["offsets"][i + 1])]
        hh, used = cm.chunk_read(pyref.calibrate(raw, CAL_RANGE, CAL_OFFSET, CAL_DIGITISATION), 4000, i + 1)
        chunk_hits.append(hit_tuple(hh)); chunk_used.append(used)
    kr = pyref.kmer_ranges()
    a, b, c, mm, ms = pyref.model_tables()
    np.savez_compressed(
        G / "ref_goldens.npz",
        ex_calibrated=sig, ex_events=ev, ex_mean_event_len=np.float32(mel), ex_total_events=np.int64(tot),
        ex_levels=levels, ex_scale=np.float32(scale), ex_shift=np.float32(shift), ex_probs=probs,
        ex_hit=hit_tuple(h), ex_hit_mel=np.float32(h.mean_event_len),
        sim_signal=sim["signal"], sim_offsets=sim["offsets"], sim_contig=sim["contig"], sim_pos=sim["pos"],
        sim_strand=sim["strand"], sim_hits=np.stack(hits), sim_mel=np.array(mels, dtype=np.float32),
        chunk_hits=np.stack(chunk_hits), chunk_used=np.array(chunk_used, dtype=np.int64),
        kmer_ranges=kr, thresholds=pyref.thresholds(), model_means=a, model_vars_x2=b, model_lognorm=c,
        model_mean=np.float32(mm), model_stdv=np.float32(ms),
        hit_fields=np.array(["mapped", "fwd", "rd_st", "rd_en", "rd_len", "rf_st", "rf_en", "rf_len", "matches",
                             "n_events", "event_i", "n_nbr", "n_sa", "n_lf"]))
    print("wrote", G / "ref_goldens.npz", "mapped", int(np.stack(hits)[:, 0].sum()), "of 48")


if __name__ == "__main__":
    main()
