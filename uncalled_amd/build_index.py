#!/usr/bin/env python3
"""BWA-format FM-index builder (numpy) -- data tooling for tests and bench.

Writes `<prefix>.{pac,ann,amb,bwt,sa}` in the on-disk layout of lh3/bwa 0.7.17 (the layout the
reference loads through bwt_restore_bwt/bwt_restore_sa/bns_restore, bwa_index.hpp:116-135, and
documented in SURVEY.md section 8c), plus `<prefix>.uncl` thresholds:

  text    T = fwd + revcomp(fwd), seq_len = 2 * l_pac (N -> lrand48()&3 after srand48(11), as bwa)
  .bwt    u64 primary; u64 L2[1..4]; then per 128 bases: 4 x u64 counts-before-block + 8 x u32
          2-bit BWT (base j of a word at bits (~j & 15) << 1), and one final counts block
  .sa     u64 primary; u64 L2[1..4]; u64 32; u64 seq_len; u64 SA[32], SA[64], ...
  .pac    forward strand, 4 bases/byte (base i at bits (~i & 3) << 1), [0x00 if l_pac%4==0], l_pac%4

The suffix array is built by prefix doubling on packed keys; fine up to a few hundred Mbp.
Self-check: `python uncalled_amd/build_index.py --selftest` rebuilds the reference's bundled example
index from example_ref.fa and compares every file byte-for-byte.
"""
import argparse
import sys
from pathlib import Path

import numpy as np

DEFAULT_UNCL = ("default\t-10.07,-3.0414736324726217,-2.531717283679984,-2.3585861097800924,"
                "-2.2706664023093497,-2.2677272727272726\t0.30176\t115.000\n")


class LRand48:
    """glibc srand48/lrand48 (48-bit LCG); bwa seeds it with 11 to replace ambiguous bases."""

    def __init__(self, seed):
        self.x = ((seed & 0xFFFFFFFF) << 16) | 0x330E

    def next(self):
        self.x = (self.x * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        return self.x >> 17


def read_fasta(path):
    names, annos, seqs = [], [], []
    cur = None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                hdr = line[1:].split(None, 1)
                names.append(hdr[0])
                annos.append(hdr[1] if len(hdr) > 1 else "")
                cur = []
                seqs.append(cur)
            elif cur is not None:
                cur.append(line)
    return names, annos, ["".join(s) for s in seqs]


_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE[ord(_c)] = _i
    _CODE[ord(_c.lower())] = _i


def encode_contigs(seqs):
    """-> (codes uint8 0..3 with N replaced, holes [(offset, len, char)] in concatenated coords, n_ambs per contig)"""
    rng = LRand48(11)
    out, holes, n_ambs = [], [], []
    offset = 0
    for s in seqs:
        raw = np.frombuffer(s.encode(), dtype=np.uint8)
        c = _CODE[raw].copy()
        amb = np.flatnonzero(c == 4)
        cnt = 0
        if amb.size:
            # runs of identical ambiguous characters become one hole each (bns_fasta2bntseq)
            brk = np.flatnonzero((np.diff(amb) != 1) | (raw[amb][1:] != raw[amb][:-1])) + 1
            starts = np.concatenate(([0], brk))
            ends = np.concatenate((brk, [amb.size]))
            for a, b in zip(starts, ends):
                holes.append((offset + int(amb[a]), int(b - a), chr(raw[amb[a]])))
                cnt += 1
            for i in amb:
                c[i] = rng.next() & 3
        n_ambs.append(cnt)
        out.append(c)
        offset += c.size
    return (np.concatenate(out) if out else np.zeros(0, np.uint8)), holes, n_ambs


def pack2(codes, per_word, dtype):
    """pack 2-bit codes MSB-first into words holding `per_word` bases (zero padded)."""
    n = codes.size
    nw = (n + per_word - 1) // per_word
    buf = np.zeros(nw * per_word, dtype=dtype)
    buf[:n] = codes
    buf = buf.reshape(nw, per_word)
    shifts = (2 * (per_word - 1 - np.arange(per_word))).astype(dtype)
    return np.bitwise_or.reduce(buf << shifts, axis=1).astype(dtype)


def suffix_array(t):
    """Suffix array of t (uint8 codes 0..3) under '$'-terminated order; returns int64[n]."""
    n = t.size
    if n == 0:
        return np.zeros(0, np.int64)
    assert n < (1 << 31), "numpy builder is limited to seq_len < 2^31"
    K0 = 21  # 3 bits per symbol (0 = past the end, 1..4 = ACGT)
    sym = np.zeros(n + K0, dtype=np.uint64)
    sym[:n] = t.astype(np.uint64) + 1
    key = np.zeros(n, dtype=np.uint64)
    for j in range(K0):
        key = (key << np.uint64(3)) | sym[j:j + n]
    del sym
    order = np.argsort(key, kind="stable")
    sk = key[order]
    del key
    newgrp = np.empty(n, dtype=bool)
    newgrp[0] = True
    np.not_equal(sk[1:], sk[:-1], out=newgrp[1:])
    del sk
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.cumsum(newgrp) - 1
    k = K0
    while True:
        ngroups = int(rank.max()) + 1
        if ngroups == n:
            break
        nxt = np.zeros(n, dtype=np.int64)  # rank of suffix i+k, shifted by 1; 0 = past the end
        if k < n:
            nxt[:n - k] = rank[k:] + 1
        key = (rank.astype(np.uint64) * np.uint64(n + 1)) + nxt.astype(np.uint64)
        del nxt
        order = np.argsort(key, kind="stable")
        sk = key[order]
        del key
        np.not_equal(sk[1:], sk[:-1], out=newgrp[1:])
        del sk
        rank[order] = np.cumsum(newgrp) - 1
        k *= 2
    sa = np.empty(n, dtype=np.int64)
    sa[rank] = np.arange(n, dtype=np.int64)
    return sa


DEFAULT_GPU_SORTER = "hip"        # tools/dev/check_sorter.py (profiles/r05_check_sorter_chr20.log): the chr20-sized build (129 M symbols) byte-identical with both sorters, 3.4 - 3.7 s vs 3.3 s


def default_sorter(dev):
    """"hip": the hand-written LSD radix sort of uncalled_amd/csrc/k_sort.hip through the C ABI (unc_sort_pairs_u64) -- what a GPU build
    uses; "torch": torch.sort (rocPRIM on the GPU; the only choice on the CPU, where the tests run this code).  UNC_INDEX_SORTER overrides."""
    import os
    want = os.environ.get("UNC_INDEX_SORTER")
    if want in ("hip", "torch"):
        return want if dev.type == "cuda" else "torch"
    return DEFAULT_GPU_SORTER if dev.type == "cuda" else "torch"


def device_argsort(key, key_bits, sorter):
    """(sorted keys, sorting permutation) of a non-negative int64 tensor, stable.  sorter "hip": in place, through unc_sort_pairs_u64
    (the permutation starts as 0 .. n - 1 inside the kernel); "torch": torch.sort."""
    import torch
    if sorter == "hip" and key.is_cuda and key.numel() > 0:
        # (round-5 advice: a library that is missing, older than the entry point, or short of device memory beside torch's cached blocks
        # must not take `uncalled index` down: torch.sort serves, with a note)
        try:
            from . import capi
            L = capi.load()
            if not hasattr(L, "unc_sort_pairs_u64"):
                raise RuntimeError("libuncalled_hip.so has no unc_sort_pairs_u64")
            order, tk, tv = torch.empty_like(key), torch.empty_like(key), torch.empty_like(key)
            args = (key.data_ptr(), order.data_ptr(), tk.data_ptr(), tv.data_ptr(), key.numel(), key_bits, True, key.device.index or 0,
                    torch.cuda.current_stream(key.device).cuda_stream, L)
            try:
                capi.sort_pairs_device(*args)
            except capi.UncalledHipError:
                # (the sort's own scratch -- the digit counts -- is a hipMalloc outside torch's caching allocator, and it is the first thing
                # the call does: the keys are untouched when it fails; once more with torch's cached blocks given back)
                torch.cuda.empty_cache()
                capi.sort_pairs_device(*args)
            return key, order
        except Exception as e:      # noqa: BLE001
            print(f"[build_index] unc_sort_pairs_u64 failed ({e!r:.200}): torch.sort instead", file=sys.stderr)
    return torch.sort(key, stable=True)


def suffix_array_torch(t, device="cuda", sorter=None):
    """Same prefix-doubling construction with the sorts on the GPU: seconds for 10^8 symbols.  The sorts are the radix sort of
    k_sort.hip (`sorter` "hip", the default on a GPU) or torch.sort; everything between them (ranks, group flags) is torch."""
    import torch
    n = int(t.size)
    assert n < (1 << 31)
    dev = torch.device(device)
    sorter = sorter or default_sorter(dev)
    K0 = 21
    sym = torch.zeros(n + K0, dtype=torch.int64, device=dev)
    sym[:n] = torch.as_tensor(t, device=dev).to(torch.int64) + 1
    key = torch.zeros(n, dtype=torch.int64, device=dev)
    for j in range(K0):
        key = (key << 3) | sym[j:j + n]
    del sym
    rank = torch.empty(n, dtype=torch.int64, device=dev)

    def rerank(key, key_bits):
        sk, order = device_argsort(key, key_bits, sorter)
        newgrp = torch.ones(n, dtype=torch.int64, device=dev)
        newgrp[1:] = (sk[1:] != sk[:-1]).to(torch.int64)
        del sk
        rank[order] = torch.cumsum(newgrp, 0) - 1
        del order, newgrp

    rerank(key, 3 * K0)
    del key
    k = K0
    while int(rank.max().item()) + 1 < n:
        nxt = torch.zeros(n, dtype=torch.int64, device=dev)
        if k < n:
            nxt[:n - k] = rank[k:] + 1
        key = rank * (n + 1) + nxt      # < 2^62 for n < 2^31
        del nxt
        rerank(key, ((n + 1) * (n + 1)).bit_length())
        del key
        k *= 2
    sa = torch.empty(n, dtype=torch.int64, device=dev)
    sa[rank] = torch.arange(n, dtype=torch.int64, device=dev)
    return sa.cpu().numpy()


def suffix_array_device(t, device="cuda", sorter=None):
    """The suffix array on a GPU.  Default: unc_build_suffix_array (the C ABI: radix sort + the steps between the sorts as HIP kernels,
    uncalled_amd/csrc/k_sort.hip) -- no torch in the process.  `sorter` "torch" / "hip", UNC_INDEX_SORTER, a library without the entry
    point, or a failure of it (device memory: 37 bytes per symbol) fall back to the torch construction above, with a note on stderr."""
    import os
    want = sorter or os.environ.get("UNC_INDEX_SORTER")
    if want in (None, "", "native") and str(device).startswith("cuda"):
        try:
            from . import capi
            L = capi.load()
            if hasattr(L, "unc_build_suffix_array"):
                dev = str(device).partition(":")[2]
                return capi.build_suffix_array(t, int(dev) if dev else 0, L)
            print("[build_index] libuncalled_hip.so has no unc_build_suffix_array: the torch construction instead", file=sys.stderr)
        except Exception as e:      # noqa: BLE001  (missing library, out of device memory, ...)
            print(f"[build_index] unc_build_suffix_array failed ({e!r:.200}): the torch construction instead", file=sys.stderr)
    return suffix_array_torch(t, device, None if want == "native" else want)


def build_from_codes(prefix, names, annos, lens, codes, holes=(), n_ambs=None, uncl_text=DEFAULT_UNCL, verbose=False, sa_device=None,
                     sorter=None):
    prefix = str(prefix)
    l_pac = int(codes.size)
    assert sum(lens) == l_pac
    n_ambs = n_ambs or [0] * len(names)

    # .pac (forward only)
    pac = pack2(codes, 4, np.uint8)
    with open(prefix + ".pac", "wb") as f:
        f.write(pac.tobytes())
        if l_pac % 4 == 0:
            f.write(b"\x00")
        f.write(bytes([l_pac % 4]))
    # .ann / .amb
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

    # text = fwd + revcomp
    t = np.concatenate((codes, (3 - codes[::-1]))).astype(np.uint8)
    n = t.size
    if verbose:
        print(f"[build_index] suffix array of {n} symbols ...", file=sys.stderr)
    sa = suffix_array_device(t, sa_device, sorter) if sa_device else suffix_array(t)
    # full matrix rows: row 0 is the sentinel suffix (SA = n)
    sa_full = np.concatenate((np.array([n], dtype=np.int64), sa))
    del sa
    primary = int(np.flatnonzero(sa_full == 0)[0])
    prev = sa_full - 1
    prev[primary] = 0
    bwt_all = t[prev]
    del prev
    bwt = np.delete(bwt_all, primary)  # the sentinel is not stored
    del bwt_all
    counts = np.bincount(t, minlength=4).astype(np.uint64)
    L2 = np.concatenate(([0], np.cumsum(counts))).astype(np.uint64)

    # interleave counts (every 128 bases) and packed words
    words = pack2(bwt, 16, np.uint32)               # ceil(n/16) words
    nblk = (n + 127) // 128
    onehot_cum = np.zeros((nblk + 1, 4), dtype=np.uint64)
    for c in range(4):
        cs = np.cumsum(bwt == c, dtype=np.uint64)
        idx = np.arange(1, nblk + 1, dtype=np.int64) * 128 - 1
        idx[-1] = min(idx[-1], n - 1)
        onehot_cum[1:, c] = cs[idx]
    # bwa writes the counts accumulated BEFORE each block, then after the last word the totals
    out = np.zeros(nblk * 16 + 8, dtype=np.uint32)
    full = np.zeros(nblk * 8, dtype=np.uint32)
    full[:words.size] = words
    blk = out[:nblk * 16].reshape(nblk, 16)
    blk[:, :8] = onehot_cum[:nblk].view(np.uint32).reshape(nblk, 8)
    blk[:, 8:] = full.reshape(nblk, 8)
    n_data_last = words.size - (nblk - 1) * 8       # data words in the last block (1..8)
    total = nblk * 16 - (8 - n_data_last)
    final_counts = onehot_cum[nblk].view(np.uint32)
    flat = np.concatenate((out[:total], final_counts))
    assert flat.size == (n + 15) // 16 + 8 * ((n + 127) // 128 + 1)
    with open(prefix + ".bwt", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes())
        f.write(L2[1:5].tobytes())
        f.write(flat.tobytes())
    # .sa
    intv = 32
    n_sa = (n + intv) // intv
    samples = sa_full[::intv][:n_sa].astype(np.uint64)
    with open(prefix + ".sa", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes())
        f.write(L2[1:5].tobytes())
        f.write(np.array([intv, n], dtype=np.uint64).tobytes())
        f.write(samples[1:].tobytes())
    if uncl_text is not None:
        with open(prefix + ".uncl", "w") as f:
            f.write(uncl_text)
    return dict(l_pac=l_pac, seq_len=n, primary=primary)


def build_from_fasta(fasta, prefix, **kw):
    names, annos, seqs = read_fasta(fasta)
    codes, holes, n_ambs = encode_contigs(seqs)
    return build_from_codes(prefix, names, annos, [len(s) for s in seqs], codes, holes, n_ambs, **kw)


def synthetic_genome(n_contigs, total_len, seed, gc=0.508, name="syn"):
    """i.i.d. ACGT contigs: returns names, lens, codes."""
    rng = np.random.default_rng(seed)
    base = total_len // n_contigs
    lens = [base] * n_contigs
    lens[-1] += total_len - base * n_contigs
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    codes = rng.choice(4, size=total_len, p=p).astype(np.uint8)
    names = [f"{name}_{i + 1}" for i in range(n_contigs)]
    return names, lens, codes


def masked_synthetic_genome(n_contigs, total_len, seed, masked_frac=0.30, mean_run=5000, gc=0.41, name="syn"):
    """SURVEY 8(d) `chr20_syn`-style reference: i.i.d. contigs with `masked_frac` of the length covered by N-runs.
    bwa replaces N by random bases and records the runs in .amb; here the runs are filled from the same seeded
    generator (the genome is synthetic anyway) and returned as holes for the .amb file."""
    names, lens, codes = synthetic_genome(n_contigs, total_len, seed, gc=gc, name=name)
    rng = np.random.default_rng(seed + 1000)
    holes, n_ambs = [], []
    off = 0
    for ln in lens:
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
    return names, lens, codes, holes, n_ambs


def write_fasta(path, names, lens, codes, width=80):
    with open(path, "w") as f:
        off = 0
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        for nm, ln in zip(names, lens):
            f.write(f">{nm}\n")
            s = lut[codes[off:off + ln]].tobytes().decode()
            for i in range(0, ln, width):
                f.write(s[i:i + width] + "\n")
            off += ln


def selftest():
    import filecmp
    import tempfile
    ref = Path("/root/reference/example")
    if not ref.exists():
        ref = Path(__file__).resolve().parents[1] / "tests/golden/example_index"
        fasta, idx = ref / "example_ref.fa", ref
    else:
        fasta, idx = ref / "example_ref.fa", ref / "index"
    with tempfile.TemporaryDirectory() as d:
        build_from_fasta(fasta, Path(d) / "x", uncl_text=None)
        ok = True
        for suf in (".pac", ".ann", ".amb", ".bwt", ".sa"):
            same = filecmp.cmp(Path(d) / ("x" + suf), idx / ("example_ref" + suf), shallow=False)
            print(f"{suf}: {'identical' if same else 'DIFFERENT'}")
            ok &= same
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--selftest", action="store_true")
    ap.add_argument("--fasta")
    ap.add_argument("--synthetic", type=int, metavar="LEN", help="total length of an i.i.d. synthetic genome")
    ap.add_argument("--contigs", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--prefix")
    a = ap.parse_args()
    if a.selftest:
        return selftest()
    if a.fasta:
        info = build_from_fasta(a.fasta, a.prefix, verbose=True)
    else:
        names, lens, codes = synthetic_genome(a.contigs, a.synthetic, a.seed)
        write_fasta(a.prefix + ".fa", names, lens, codes)
        info = build_from_codes(a.prefix, names, [""] * len(names), lens, codes, verbose=True)
    print(info)
    return 0


if __name__ == "__main__":
    sys.exit(main())
