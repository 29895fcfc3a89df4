"""`uncalled pafstats`: PAF parsing, accuracy against a reference PAF and speed summary.

Mirrors the behaviour (class and function names, classification rules, printed report) of the reference's
uncalled/pafstats.py:8-233; tests/test_pafstats.py pins the report text against fixtures produced by running that file.
"""
import sys

import numpy as np

_STR_TAG_TYPES = ("A", "Z", "B", "H")


class PafEntry:
    """One PAF record (pafstats.py:8-105).  Built from a text line, or from a 12-item list when reversing."""

    def __init__(self, line, tags=None):
        from_text = not isinstance(line, list)
        cols = line.split() if from_text else line
        self.qr_name = cols[0]
        self.qr_len = int(cols[1])
        self.is_mapped = cols[4] != ("*" if from_text else None)
        if self.is_mapped:
            self.qr_st, self.qr_en = int(cols[2]), int(cols[3])
            self.is_fwd = cols[4] == ("+" if from_text else True)
            self.rf_name = cols[5]
            self.rf_len, self.rf_st, self.rf_en = int(cols[6]), int(cols[7]), int(cols[8])
            self.match_num, self.aln_len, self.qual = int(cols[9]), int(cols[10]), int(cols[11])
        else:
            self.qr_st, self.qr_en = 1, self.qr_len
            self.is_fwd = self.rf_name = self.rf_len = self.rf_st = self.rf_en = None
            self.match_num = self.aln_len = self.qual = None
        self.tags = {} if tags is None else tags
        for field in cols[12:]:
            key, typ, val = field.split(":")
            if typ == "f":
                val = float(val)
            elif typ == "i":
                val = int(val)
            elif typ not in _STR_TAG_TYPES:
                sys.stderr.write("Error: invalid tag type \"%s\"\n" % typ)
                sys.exit(1)
            self.tags[key] = (val, typ)

    def rev(self):
        return PafEntry([self.rf_name, self.rf_len, self.rf_st, self.rf_en, self.is_fwd, self.qr_name, self.qr_len,
                         self.qr_st, self.qr_en, self.match_num, self.aln_len, self.qual], self.tags)

    def get_tag(self, key):
        return self.tags.get(key, (None, None))[0]

    def set_tag(self, key, val, typ=None):
        if typ is None:
            typ = "i" if isinstance(val, int) else "f" if isinstance(val, float) else "Z"
        self.tags[key] = (val, typ)

    def qry_loc(self):
        return (self.qr_name, self.qr_st, self.qr_en)

    def ref_loc(self):
        return (self.rf_name, self.rf_st, self.rf_en)

    def ext_ref(self, ext=1.0):
        """Reference span widened by the unaligned read ends, scaled by `ext`."""
        head = int(self.qr_st * ext)
        tail = int((self.qr_len - self.qr_en) * ext)
        lo, hi = (head, tail) if self.is_fwd else (tail, head)
        return (max(1, self.rf_st - lo), min(self.rf_len, self.rf_en + hi))

    def _same_ref(self, other):
        return self.is_mapped and other.is_mapped and self.rf_name.startswith(other.rf_name)

    def overlaps(self, other, ext=0.0):
        if not self._same_ref(other):
            return False
        a = self.ext_ref(ext)
        b = other.ext_ref(ext)
        return max(a[0], b[0]) <= min(a[1], b[1])

    def contains(self, other):
        return self._same_ref(other) and self.rf_st <= other.rf_st and self.rf_en >= other.rf_en

    def __lt__(self, other):
        return self.qr_name < self.qr_name   # as in the reference: never true

    def __str__(self):
        tagstr = "\t".join("%s:%s:%s" % (k, t, v) for k, (v, t) in self.tags.items())
        if not self.is_mapped:
            return "\t".join([self.qr_name, str(self.qr_len)] + ["*"] * 10 + [tagstr])
        return "%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%s" % (
            self.qr_name, self.qr_len, self.qr_st, self.qr_en, "+" if self.is_fwd else "-", self.rf_name, self.rf_len,
            self.rf_st, self.rf_en, self.match_num, self.aln_len, self.qual, tagstr)


def parse_paf(infile, max_load=None):
    if isinstance(infile, str):
        infile = open(infile)
    n = 0
    for line in infile:
        if line[0] == "#":
            continue
        if max_load is not None and n >= max_load:
            break
        yield PafEntry(line)
        n += 1


def paf_ref_compare(qry, ref, ret_qry=True, check_locs=True, ext=1.5):
    """Classify query records against truth records of the same read: (tp, tn, fp, fn, fp_unmap)."""
    if isinstance(ref, dict):
        truth = ref
    else:
        truth = {}
        for r in ref:
            truth.setdefault(r.qr_name, []).append(r)
    tp, tn, fp, fn, fp_unmap = [], [], [], [], []
    for q in qry:
        rs = truth.get(q.qr_name, [None])
        truth_unmapped = rs == [None] or not rs[0].is_mapped
        if not q.is_mapped:
            (tn if truth_unmapped else fn).append(q if ret_qry else rs[0])
            continue
        if truth_unmapped:
            fp_unmap.append(q if ret_qry else rs[0])
            continue
        hit = None
        for r in rs:
            if (q.overlaps(r, ext) if check_locs else q.rf_name == r.rf_name):
                hit = r
                break
        if hit is not None:
            tp.append(q if ret_qry else hit)
        else:
            fp.append(q if ret_qry else rs[-1])
    return tp, tn, fp, fn, fp_unmap


def add_opts(parser):
    parser.add_argument("infile", type=str, help="PAF file output by UNCALLED")
    parser.add_argument("-n", "--max-reads", required=False, type=int, default=None,
                        help="Will only look at first n reads if specified")
    parser.add_argument("-r", "--ref-paf", required=False, type=str, default=None,
                        help="Reference PAF file. Will output percent true/false positives/negatives with respect to "
                             "reference. Reads not mapped in reference PAF will be classified as NA.")
    parser.add_argument("-a", "--annotate", action="store_true",
                        help="Should be used with --ref-paf. Will output an annotated version of the input with T/P F/P "
                             "specified in an 'rf' tag")


def run(args, out=None, err=None):
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    locs = list(parse_paf(args.infile, args.max_reads))
    n = len(locs)
    n_mapped = sum(p.is_mapped for p in locs)
    stats = err if args.annotate else out
    stats.write("Summary: %d reads, %d mapped (%.2f%%)\n\n" % (n, n_mapped, 100 * n_mapped / n))

    if args.ref_paf is not None:
        stats.write("Comparing to reference PAF\n")
        groups = paf_ref_compare(locs, parse_paf(args.ref_paf))
        ntp, ntn, nfp, nfn, nna = map(len, groups)
        stats.write("     P     N\n")
        stats.write("T %6.2f %5.2f\n" % (100 * ntp / n, 100 * ntn / n))
        stats.write("F %6.2f %5.2f\n" % (100 * nfp / n, 100 * nfn / n))
        stats.write("NA: %.2f\n\n" % (100 * nna / n))
        if args.annotate:
            for grp, label in zip(groups, ("tp", "tn", "fp", "fn", "na")):
                for p in grp:
                    p.set_tag("rf", label, "Z")
                    out.write("%s\n" % p)

    if locs[0].get_tag("mt") is not None:
        ms = np.array([p.get_tag("mt") for p in locs if p.is_mapped])
        bp = np.array([p.qr_en for p in locs if p.is_mapped])
        rate = 1000 * bp / ms
        stats.write("Speed            Mean    Median\n")
        stats.write("BP per sec: %9.2f %9.2f\n" % (np.mean(rate), np.median(rate)))
        stats.write("BP mapped:  %9.2f %9.2f\n" % (np.mean(bp), np.median(bp)))
        stats.write("MS to map:  %9.2f %9.2f\n" % (np.mean(ms), np.median(ms)))
