#!/usr/bin/env python3
"""Seeded r9.4.1 raw-signal read simulator -- data tooling for tests/bench.

For each read: pick contig / position / strand uniformly, `read_bases` bases (default 3600 = about 8 s =
32 k samples); every 5-mer of the read (5'->3') emits one level ~ N(mu_k, sigma_k) from the r9.4 template
model, held for dwell ~ Geometric(mean 8.9) samples, plus per-sample N(0, 1.5 pA) noise; per-read
scale ~ N(1, 0.05) and shift ~ N(0, 5); `off_target` of the reads are random sequence; quantised to int16 with
range=1534.14, digitisation=8192, offset=10 (the example fast5's calibration).
"""
import struct
from pathlib import Path

import numpy as np

CAL_RANGE, CAL_DIGITISATION, CAL_OFFSET = 1534.14, 8192.0, 10.0


def load_template_model():
    """(means[1024], stdvs[1024]) float32, template orientation, from the generated bit table."""
    txt = (Path(__file__).resolve().parents[1] / "uncalled_amd/csrc/r94_model_table.h").read_text()
    import re
    bits = [int(x, 16) for x in re.findall(r"0x([0-9A-F]{8})u", txt)]
    assert len(bits) == 2048
    vals = np.array(struct.unpack("<2048f", struct.pack("<2048I", *bits)), dtype=np.float32)
    return vals[0::2].copy(), vals[1::2].copy()


def simulate_reads(codes, contig_lens, n_reads, seed=42, read_bases=3600, off_target=0.10, dwell_mean=8.9,
                   noise_sd=1.5, chunk=1024, model=None):
    """-> dict(signal int16[total], offsets uint64[n+1], contig int32[n] (-1 = off target), pos int64[n], strand int8[n])"""
    means, stdvs = model if model is not None else load_template_model()
    rng = np.random.default_rng(seed)
    contig_lens = np.asarray(contig_lens, dtype=np.int64)
    contig_off = np.concatenate(([0], np.cumsum(contig_lens)))
    usable = np.maximum(contig_lens - read_bases, 0)
    assert usable.sum() > 0 or off_target >= 1.0, "contigs shorter than the read length"
    pw = usable / usable.sum() if usable.sum() > 0 else None
    sig_parts, lens = [], []
    contig = np.full(n_reads, -1, dtype=np.int32)
    pos = np.zeros(n_reads, dtype=np.int64)
    strand = np.zeros(n_reads, dtype=np.int8)
    nk = read_bases - 4
    ar = np.arange(read_bases, dtype=np.int64)
    for r0 in range(0, n_reads, chunk):
        r1 = min(n_reads, r0 + chunk)
        R = r1 - r0
        off = rng.random(R) < off_target
        seqs = np.empty((R, read_bases), dtype=np.uint8)
        if pw is not None:
            cg = rng.choice(len(contig_lens), size=R, p=pw)
            ps = (rng.random(R) * (usable[cg] + 1)).astype(np.int64)
            st = rng.integers(0, 2, size=R).astype(np.int8)  # 0 = '+', 1 = '-'
            g = codes[(contig_off[cg] + ps)[:, None] + ar[None, :]]
            rc = 3 - g[:, ::-1]
            seqs[:] = np.where(st[:, None] == 0, g, rc)
            contig[r0:r1] = np.where(off, -1, cg)
            pos[r0:r1] = np.where(off, 0, ps)
            strand[r0:r1] = np.where(off, 0, st)
        n_off = int(off.sum())
        if n_off:
            seqs[off] = rng.integers(0, 4, size=(n_off, read_bases), dtype=np.uint8)
        k = seqs[:, 0:nk].astype(np.int32)
        for j in range(1, 5):
            k = (k << 2) | seqs[:, j:j + nk]
        lv = means[k] + stdvs[k] * rng.standard_normal((R, nk), dtype=np.float32)
        scale = rng.normal(1.0, 0.05, size=R).astype(np.float32)
        shift = rng.normal(0.0, 5.0, size=R).astype(np.float32)
        lv = lv * scale[:, None] + shift[:, None]
        dwell = rng.geometric(1.0 / dwell_mean, size=(R, nk)).astype(np.int64)
        rl = dwell.sum(axis=1)
        pa = np.repeat(lv.ravel(), dwell.ravel())
        pa += noise_sd * rng.standard_normal(pa.size, dtype=np.float32)
        raw = np.rint(pa * np.float32(CAL_DIGITISATION / CAL_RANGE)) - np.float32(CAL_OFFSET)
        sig_parts.append(np.clip(raw, 0, 32767).astype(np.int16))
        lens.append(rl)
    lens = np.concatenate(lens) if lens else np.zeros(0, np.int64)
    offsets = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
    signal = np.concatenate(sig_parts) if sig_parts else np.zeros(0, np.int16)
    return dict(signal=signal, offsets=offsets, contig=contig, pos=pos, strand=strand)


if __name__ == "__main__":
    import argparse
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from uncalled_amd.build_index import read_fasta, encode_contigs
    ap = argparse.ArgumentParser()
    ap.add_argument("fasta")
    ap.add_argument("out_npz")
    ap.add_argument("-n", type=int, default=100)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--bases", type=int, default=3600)
    a = ap.parse_args()
    names, _, seqs = read_fasta(a.fasta)
    codes, _, _ = encode_contigs(seqs)
    d = simulate_reads(codes, [len(s) for s in seqs], a.n, seed=a.seed, read_bases=a.bases)
    np.savez_compressed(a.out_npz, **d)
    print(a.out_npz, d["offsets"].size - 1, "reads", d["signal"].size, "samples")
