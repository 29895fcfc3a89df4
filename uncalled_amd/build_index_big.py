#!/usr/bin/env python3
"""BWA-format index builder for references past the 2^31-symbol limit of uncalled_amd/build_index.py (GRCh38: 6.2 G symbols).

Same files, byte for byte, as `bwa index` / uncalled_amd/build_index.py (tests/test_build_index_big.py compares them on the
bundled example and on repeat-rich synthetic genomes); the difference is how the suffix array comes about:

  * suffixes are bucketed by their first 8 symbols (3 bits each: 0 = past the end, 1..4 = ACGT; 2^24 buckets whose order
    is the lexicographic one), consecutive buckets are grouped into chunks of at most `chunk` suffixes, and every chunk
    is a contiguous range of the final suffix array;
  * a chunk is sorted on its own: 63-bit key of the first 21 symbols, one sort, then the groups of equal keys (repeats)
    are refined 21 symbols at a time, only the tied suffixes taking part;
  * the BWT symbol and the sampled SA entry of every row are emitted chunk by chunk, so nothing of size n x 8 bytes is
    ever resident: the text (1 byte per symbol), the BWT (1 byte per symbol) and one chunk's keys / positions.

torch does the sorting (device "cuda" on the GPU box, "cpu" in the tests).  Working set for GRCh38 (n = 6.2e9) with the
default chunk of 2^28: 6.2 GB text + 6.2 GB BWT + about 12 GB per chunk.
"""
import os
import sys
from pathlib import Path

import numpy as np

from .build_index import DEFAULT_UNCL, default_sorter

K = 21            # symbols per 63-bit key
B = 8             # symbols of the bucket id


def _sym_at(t, pos, n):
    """3-bit symbol (0 past the end, 1..4) of text positions `pos` (int64 tensor, may reach past n)."""
    import torch
    inside = pos < n
    v = t[torch.where(inside, pos, torch.zeros_like(pos))].to(torch.int64) + 1
    return torch.where(inside, v, torch.zeros_like(v))


def _key(t, pos, n, depth, width=K):
    """big-endian 3-bit packing of the `width` symbols starting `depth` symbols into the suffixes at `pos`"""
    import torch
    key = torch.zeros_like(pos)
    for j in range(width):
        key = (key << 3) | _sym_at(t, pos + (depth + j), n)
    return key


def _sort_chunk(t, pos, n, sorter="torch"):
    """positions of one chunk -> the same positions in suffix order.  sorter: build_index.device_argsort's ("hip": the radix sort of
    k_sort.hip, the default on a GPU: grch38_syn comes out byte-identical in 61.9 s against 61.0 s with torch.sort,
    profiles/r05_check_sorter_grch38.log)"""
    import torch
    from .build_index import device_argsort
    key = _key(t, pos, n, 0)
    key, order = device_argsort(key, 3 * K, sorter)
    pos = pos[order]
    del order
    m = pos.numel()
    # group id = index of the first element of the run of equal keys
    newgrp = torch.ones(m, dtype=torch.bool, device=pos.device)
    newgrp[1:] = key[1:] != key[:-1]
    del key
    depth = K
    while True:
        # a suffix is tied when its run has more than one element
        start = torch.nonzero(newgrp).flatten()
        size = torch.diff(torch.cat((start, torch.tensor([m], device=pos.device))))
        grp = torch.cumsum(newgrp.to(torch.int64), 0) - 1                  # run index per element
        tied = (size > 1)[grp]
        idx = torch.nonzero(tied).flatten()
        if idx.numel() == 0:
            return pos
        sub_pos = pos[idx]
        sub_grp = grp[idx]
        del grp, tied, size, start
        k2 = _key(t, sub_pos, n, depth)
        # order by (run, next key): stable sort by the key, then by the run
        k2s, o1 = device_argsort(k2, 3 * K, sorter)
        g1 = sub_grp[o1]
        g2, o2 = device_argsort(g1, max(1, int(m).bit_length()), sorter)
        perm = o1[o2]
        k2f = k2s[o2]
        pos[idx] = sub_pos[perm]          # the tied slots of a run are contiguous and ascending: refill in the new order
        brk = torch.ones(idx.numel(), dtype=torch.bool, device=pos.device)
        brk[1:] = (g2[1:] != g2[:-1]) | (k2f[1:] != k2f[:-1])
        newgrp[idx] = brk
        depth += K
        del sub_pos, sub_grp, k2, k2s, o1, g1, g2, o2, perm, k2f, brk, idx


def suffix_rows(t_in, device="cuda", chunk=1 << 28, piece=1 << 27, verbose=False, sorter=None):
    """Generator over the suffix array of t (uint8 codes 0..3 -- ndarray or tensor --, '$'-terminated order) in order, one
    chunk (int64 tensor of text positions, on `device`) at a time; the text tensor comes along with every item."""
    import torch
    dev = torch.device(device)
    t = torch.as_tensor(t_in, device=dev)
    n = int(t.numel())
    nb = 1 << (3 * B)
    # histogram of bucket ids, text scanned in pieces
    hist = torch.zeros(nb, dtype=torch.int64, device=dev)

    def buckets(lo, hi):
        p = torch.arange(lo, hi, dtype=torch.int64, device=dev)
        return _key(t, p, n, 0, B)

    for lo in range(0, n, piece):
        hist += torch.bincount(buckets(lo, min(n, lo + piece)), minlength=nb)
    csum = torch.cumsum(hist, 0).cpu().numpy()
    del hist
    # consecutive buckets -> chunks of about `chunk` suffixes: cut where the running count passes a multiple of `chunk`
    # (a chunk exceeds it by at most one bucket; a bucket larger than `chunk` is a chunk of its own)
    cuts = np.unique(np.searchsorted(csum, np.arange(chunk, n, chunk, dtype=np.int64), side="right"))
    edges = [0] + [int(c) for c in cuts if 0 < c < nb] + [nb]
    bounds = [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]
    done = 0
    for ci, (blo, bhi) in enumerate(bounds):
        parts = []
        for lo in range(0, n, piece):
            hi = min(n, lo + piece)
            bk = buckets(lo, hi)
            sel = torch.nonzero((bk >= blo) & (bk < bhi)).flatten() + lo
            if sel.numel():
                parts.append(sel)
        pos = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
        del parts
        if verbose:
            print(f"[build_index_big] chunk {ci + 1}/{len(bounds)}: {pos.numel()} suffixes", file=sys.stderr, flush=True)
        pos = _sort_chunk(t, pos, n, sorter or default_sorter(dev))
        yield t, pos
        done += pos.numel()
    assert done == n, (done, n)


def big_masked_genome(n_contigs, total_len, seed, masked_frac=0.30, mean_run=5000, name="syn"):
    """`grch38_syn`-style reference: i.i.d. contigs, `masked_frac` of the length recorded as N-runs (bwa
    fills N with random bases, the genome is random anyway: the runs only go to .amb).  Generated contig by contig from
    one random byte per base (76/52/52/76 of 256 values for A/C/G/T: GC 0.406), a few seconds for 3.1 Gbp."""
    base = total_len // n_contigs
    lens = [base] * n_contigs
    lens[-1] += total_len - base * n_contigs
    lut = np.empty(256, dtype=np.uint8)
    lut[:76], lut[76:128], lut[128:180], lut[180:] = 0, 1, 2, 3
    codes = np.empty(total_len, dtype=np.uint8)
    holes, n_ambs, off = [], [], 0
    for i, ln in enumerate(lens):
        rng = np.random.default_rng([seed, i])
        np.take(lut, rng.integers(0, 256, size=ln, dtype=np.uint8), out=codes[off:off + ln])
        n_runs = max(1, int(ln * masked_frac / mean_run))
        starts = np.sort(rng.integers(0, max(1, ln - mean_run), n_runs))
        runs = rng.geometric(1.0 / mean_run, n_runs)
        cnt, last_end = 0, 0
        for st, rl in zip(starts, runs):
            st = max(int(st), last_end + 1)
            en = min(ln, st + int(rl))
            if en <= st:
                continue
            holes.append((off + st, en - st, "N"))
            last_end = en
            cnt += 1
        n_ambs.append(cnt)
        off += ln
    names = [f"{name}_{i + 1}" for i in range(n_contigs)]
    return names, lens, codes, holes, n_ambs


def build_from_codes_big(prefix, names, annos, lens, codes, holes=(), n_ambs=None, uncl_text=DEFAULT_UNCL, device="cuda",
                         chunk=1 << 28, piece=1 << 27, verbose=False, sorter=None):
    """Drop-in for build_index.build_from_codes with the chunked suffix sort; writes .pac .ann .amb .bwt .sa (.uncl)."""
    import torch
    prefix = str(prefix)
    l_pac = int(codes.size)
    assert sum(lens) == l_pac
    n_ambs = n_ambs or [0] * len(names)
    dev = torch.device(device)
    codes_t = torch.as_tensor(codes).to(dev)
    with open(prefix + ".pac", "wb") as f:      # 4 bases per byte, first base in the top bits; packed on the device in pieces
        step4 = piece * 4
        for lo in range(0, l_pac, step4):
            seg = codes_t[lo:min(l_pac, lo + step4)]
            pad = (-seg.numel()) % 4
            if pad:
                seg = torch.cat((seg, torch.zeros(pad, dtype=torch.uint8, device=dev)))
            q = seg.view(-1, 4)
            f.write(((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).cpu().numpy().tobytes())
        if l_pac % 4 == 0:
            f.write(b"\x00")
        f.write(bytes([l_pac % 4]))
    with open(prefix + ".ann", "w") as f:
        f.write(f"{l_pac} {len(names)} 11\n")
        off = 0
        for nm, an, ln, na in zip(names, annos, lens, n_ambs):
            f.write(f"0 {nm} {an if an else '(null)'}\n")
            f.write(f"{off} {ln} {na}\n")
            off += ln
    with open(prefix + ".amb", "w") as f:
        f.write(f"{l_pac} {len(names)} {len(holes)}\n")
        for off, ln, ch in holes:
            f.write(f"{off} {ln} {ch}\n")

    t_all = torch.cat((codes_t, 3 - codes_t.flip(0)))      # forward strand + reverse complement, as bwa builds it
    del codes_t
    n = int(t_all.numel())
    intv = 32
    n_sa = (n + intv) // intv
    samples = np.zeros(n_sa, dtype=np.uint64)          # SA of the rows r = 0, 32, 64, .. of the matrix with the sentinel row
    samples[0] = n                                      # row 0 is the sentinel suffix (never written: bwa stores from row 32 on)
    bwt = torch.zeros(n, dtype=torch.uint8, device=dev)  # stored BWT: the row of suffix 0 (primary) is left out
    primary = None
    row = 1                                             # matrix row of the next suffix (row 0 = sentinel)
    t = None
    for t, pos in suffix_rows(t_all, device, chunk, piece, verbose, sorter):
        m = pos.numel()
        rows = torch.arange(row, row + m, dtype=torch.int64, device=dev)
        zero = torch.nonzero(pos == 0).flatten()
        if zero.numel():
            primary = int(rows[zero[0]].item())
        sym = t[torch.where(pos > 0, pos - 1, torch.zeros_like(pos))]
        # stored index: rows before the primary keep their number, rows after it move up by one; the primary row is dropped
        keep = pos != 0
        if primary is None:
            dst = rows
        else:
            dst = torch.where(rows > primary, rows - 1, rows)
        bwt[dst[keep]] = sym[keep]
        smp = torch.nonzero(rows % intv == 0).flatten()
        if smp.numel():
            samples[(rows[smp] // intv).cpu().numpy()] = pos[smp].cpu().numpy().astype(np.uint64)
        row += m
        del rows, sym, keep, dst, smp, zero
    assert primary is not None and row == n + 1
    bwt[0] = t[n - 1]                                   # row 0 (the sentinel suffix) is preceded by the last symbol
    counts = torch.bincount(t.to(torch.int64), minlength=4).cpu().numpy().astype(np.uint64)
    L2 = np.concatenate(([0], np.cumsum(counts))).astype(np.uint64)

    # .bwt: per 128-symbol block the counts BEFORE the block (4 x u64) then 8 x u32 of 2-bit symbols; totals at the end
    nblk = (n + 127) // 128
    with open(prefix + ".bwt", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes())
        f.write(L2[1:5].tobytes())
        run = np.zeros(4, dtype=np.uint64)
        step = max(1, (piece // 128)) * 128
        n_words = (n + 15) // 16
        words_left = n_words
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            seg = bwt[lo:hi]
            nb_seg = (hi - lo + 127) // 128
            pad = nb_seg * 128 - (hi - lo)
            if pad:
                seg = torch.cat((seg, torch.zeros(pad, dtype=torch.uint8, device=dev)))
            blk = seg.view(nb_seg, 128)
            valid = None
            if pad:
                valid = torch.ones(nb_seg * 128, dtype=torch.bool, device=dev)
                valid[hi - lo:] = False
                valid = valid.view(nb_seg, 128)
            cnt = torch.zeros((nb_seg, 4), dtype=torch.int64, device=dev)
            for c in range(4):
                eq = blk == c
                if valid is not None:
                    eq = eq & valid
                cnt[:, c] = eq.sum(1)
            before = torch.cumsum(cnt, 0) - cnt
            before_np = before.cpu().numpy().astype(np.uint64) + run
            run = run + cnt.sum(0).cpu().numpy().astype(np.uint64)
            sh = (2 * (15 - torch.arange(16, device=dev))).to(torch.int64)
            w = (blk.view(nb_seg, 8, 16).to(torch.int64) << sh).sum(2).to(torch.int64)
            w_np = w.cpu().numpy().astype(np.uint32)
            out = np.zeros((nb_seg, 16), dtype=np.uint32)
            out[:, :8] = before_np.view(np.uint32).reshape(nb_seg, 8)
            out[:, 8:] = w_np
            flat = out.reshape(-1)
            # the last block of the file carries only the words that hold data
            seg_words = min(words_left, nb_seg * 8)
            if seg_words < nb_seg * 8:
                flat = flat[:(nb_seg - 1) * 16 + 8 + (seg_words - (nb_seg - 1) * 8)]
            words_left -= seg_words
            f.write(flat.tobytes())
            del seg, blk, cnt, before, w
        f.write(run.astype(np.uint64).view(np.uint32).tobytes())
    with open(prefix + ".sa", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes())
        f.write(L2[1:5].tobytes())
        f.write(np.array([intv, n], dtype=np.uint64).tobytes())
        f.write(samples[1:].tobytes())
    if uncl_text is not None:
        with open(prefix + ".uncl", "w") as f:
            f.write(uncl_text)
    return dict(l_pac=l_pac, seq_len=n, primary=primary)
