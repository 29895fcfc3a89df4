"""`uncalled index` parameter search: FM-range-size trajectories of a self-aligned reference -> per-range-size
event-probability thresholds, written as `<prefix>.uncl` (the file Mapper::load_static parses, mapper.cpp:123-157).

Host mirror of the reference's IndexParameterizer (uncalled/index.py:53-209) on top of the device self-alignment
(`unc_self_align`, which replaces src/self_align_ref.cpp:34-91).  The arithmetic follows the reference step by step
(same numpy reductions in the same order) so that the bundled example index's `.uncl` line is reproduced verbatim.
Model table: uncalled/conf/r94_5mers_threshs.txt (threshold, fraction of events matching, mean k-mers matched),
stored as uncalled_amd/data/r94_5mers_threshs.npz.
"""
from pathlib import Path

import numpy as np

DATA = Path(__file__).resolve().parent / "data" / "r94_5mers_threshs.npz"

# defaults of `uncalled index` (uncalled/args.py:86-140)
DEFAULTS = dict(max_sample_dist=100, min_samples=50000, max_samples=1000000, kmer_len=5, matchpr1=0.6334, matchpr2=0.9838,
                pathlen_percentile=0.05, max_replen=100)


def choose_sample_dist(ref_len, max_sample_dist=100, min_samples=50000, max_samples=1000000):
    """index.py:74-81: sampling distance from the packed reference length (first field of the .ann header)."""
    approx = ref_len / max_sample_dist
    if approx < min_samples:
        return int(np.ceil(ref_len / min_samples))
    if approx > max_samples:
        return int(np.floor(ref_len / max_samples))
    return max_sample_dist


def _ramp(xmax, ymin, ymax, exp, n=100):
    """index.py:47-51 (power_fn)"""
    dt = 1.0 / n
    t = np.arange(0, 1 + dt, dt)
    return t * xmax, (t ** exp) * (ymax - ymin) + ymin


class IndexParameterizer:
    def __init__(self, lens, full_len, kmer_len=5, matchpr1=0.6334, matchpr2=0.9838, pathlen_percentile=0.05, max_replen=100):
        """lens: uint64[n, cap] range sizes per trajectory (zero padded), full_len: true trajectory lengths."""
        self.pck1, self.pck2 = matchpr1, matchpr2
        self._map_stats(np.asarray(lens), np.asarray(full_len, dtype=np.int64), kmer_len, pathlen_percentile, max_replen)
        tab = np.load(DATA)
        # index.py:120-134: stored best-threshold-first, used reversed
        self.model_ekms = np.flip(tab["thresh"], 0)
        self.model_pcks = np.flip(tab["freq"], 0)
        self.model_counts = np.flip(tab["count"], 0)
        self.presets = {}

    def _map_stats(self, lens, full_len, k, percentile, max_replen):
        """index.py:66-118 (calc_map_stats).  A trajectory shorter than k counts as the single entry [1] (:84)."""
        n, cap = lens.shape
        short = full_len < k
        klen = np.where(short, 1, full_len - (k - 1))                 # len(p[k-1:])
        first = np.where(short, 1, lens[:, k - 1] if cap >= k else 1)   # p[k-1]
        kept = klen[klen <= max_replen]
        longest = int(kept.max())
        gt = np.array([(kept > i).sum() for i in range(longest)], dtype=np.float64)   # gt1_counts, :88-91
        max_pathlen = int(np.flatnonzero(gt / len(kept) <= percentile)[0])
        assert k - 1 + max_pathlen <= cap, "trajectory head buffer too short"
        max_fmexp = int(np.log2(int(first.max()))) + 1
        mat = np.zeros((max_fmexp, max_pathlen))
        for i in range(max_pathlen):                                   # :96-100, column by column
            live = (~short) & (klen > i)
            vals = lens[live, k - 1 + i]
            if i == 0:
                live0 = short                                          # the [1] stand-ins sit in column 0, row log2(1) = 0
                mat[0, 0] += int(live0.sum())
            rows = np.log2(vals.astype(np.float64)).astype(np.int64)
            np.add.at(mat[:, i], rows, 1)
            dead = n - int(live.sum()) - (int(short.sum()) if i == 0 else 0)
            mat[0, i] += dead                                          # ended trajectories count as size 1
        self.fm_path_mat = mat
        pos = np.arange(max_pathlen)
        # (an empty row is 0 / 0 = nan in the reference as well, index.py:103-106, where numpy says so on stderr in every run: the same
        # value, without the RuntimeWarning)
        with np.errstate(invalid="ignore", divide="ignore"):
            self.fm_locs = np.array([np.sum(np.array([i * w for i, w in zip(pos, mat[f] / np.sum(mat[f]))])) for f in range(max_fmexp)])
            exps = np.arange(max_fmexp)
            self.loc_fms = np.array([np.sum(np.array([i * w for i, w in zip(exps, mat[:, p] / np.sum(mat[:, p]))])) for p in range(max_pathlen)])
        self.speed_denom = np.sum(self.loc_fms)
        self.conf_locs = np.arange(np.round(self.fm_locs[0]))
        self.all_locs = np.arange(max_pathlen)

    def fn_speed(self, locs, pcks):
        """index.py:136-140"""
        p = np.interp(self.all_locs, locs, pcks)
        counts = np.interp(p, self.model_pcks, self.model_counts)
        return np.dot(counts, self.loc_fms) / self.speed_denom

    def fn_prob(self, locs, pcks):
        """index.py:142-143"""
        return np.prod(np.interp(self.conf_locs, locs, pcks))

    def add_preset(self, name, tgt_prob=None, tgt_speed=None, exp_st=2, init_fac=2, eps=0.00001):
        """index.py:145-200: bisection on the exponent of the match-probability ramp."""
        exp, lo, hi, prev = exp_st, None, None, None
        while True:
            locs, pcks = _ramp(self.fm_locs[0], self.pck1, self.pck2, exp)
            delta = (self.fn_prob(locs, pcks) - tgt_prob) if tgt_prob is not None else (self.fn_speed(locs, pcks) - tgt_speed)
            if abs(delta) <= eps or delta == prev:
                break
            prev = delta
            if delta < 0:
                hi = exp
            else:
                lo = exp
            last = exp
            if hi is None:
                exp *= init_fac
            elif lo is None:
                exp /= init_fac
            else:
                exp = lo + ((hi - lo) / 2.0)
            if exp == last:
                break
        fm_pcks = np.interp(self.fm_locs, locs, pcks)
        ekms = np.interp(fm_pcks, self.model_pcks, self.model_ekms)
        self.presets[name] = (ekms, self.fn_prob(locs, pcks), self.fn_speed(locs, pcks))
        return self.presets[name]

    def text(self):
        """index.py:202-209"""
        return "".join("%s\t%s\t%.5f\t%.3f\n" % (name, ",".join(map(str, ekms)), prob, speed)
                       for name, (ekms, prob, speed) in self.presets.items())


def parameterize(index, prefix, presets=(("default", dict(tgt_speed=115)),), write=True, **opts):
    """scripts/uncalled:38-78 (`uncalled index` after the BWA index exists): self-align on the GPU, search, write .uncl."""
    o = dict(DEFAULTS)
    o.update(opts)
    with open(str(prefix) + ".ann") as f:
        ref_len = int(f.readline().split()[0])
    dist = choose_sample_dist(ref_len, o["max_sample_dist"], o["min_samples"], o["max_samples"])
    lens, full = index.self_align(prefix, dist, cap=128)
    p = IndexParameterizer(lens, full, o["kmer_len"], o["matchpr1"], o["matchpr2"], o["pathlen_percentile"], o["max_replen"])
    for name, kw in presets:
        p.add_preset(name, **kw)
    if write:
        Path(str(prefix) + ".uncl").write_text(p.text())
    return p
