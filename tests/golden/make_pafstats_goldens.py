"""Generates tests/golden/pafstats/*: two seeded PAF files and the reports the REFERENCE's uncalled/pafstats.py prints for
them (plain, --ref-paf, --ref-paf --annotate).  Run in the build container only (needs /root/reference)."""
import argparse
import contextlib
import importlib.util
import io
import random
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE / "pafstats"


def make_pafs(seed=7, n=60):
    rnd = random.Random(seed)
    refs = [("chrA", 500000), ("chrB", 250000), ("chrA_alt", 90000)]
    qry, truth = [], []
    for i in range(n):
        name = "read%03d" % i
        rd_len = rnd.randint(2000, 30000)
        rf, rf_len = rnd.choice(refs)
        st = rnd.randint(1, rf_len - rd_len - 1)
        fwd = rnd.random() < 0.5
        t_mapped = rnd.random() < 0.8
        if t_mapped:
            truth.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t60\ttp:A:P" % (name, rd_len, 10, rd_len - 20, "+" if fwd else "-", rf, rf_len, st,
                                                                                  st + rd_len, rd_len // 2, rd_len))
            if rnd.random() < 0.15:   # secondary truth record on another contig
                truth.append("%s\t%d\t%d\t%d\t+\tchrB\t250000\t%d\t%d\t%d\t%d\t0" % (name, rd_len, 5, 900, 1000, 1900, 400, 900))
        else:
            truth.append("%s\t%d\t*\t*\t*\t*\t*\t*\t*\t*\t*\t255" % (name, rd_len))
        mt = rnd.uniform(5, 400)
        tags = "ch:i:%d\tst:i:%d\tmt:f:%.6f" % (rnd.randint(1, 512), rnd.randint(0, 10 ** 7), mt)
        u = rnd.random()
        if u < 0.65:
            q_rf, q_len = rf, rf_len
            q_fwd = fwd
            qs = rnd.randint(50, 600)
            qe = qs + rnd.randint(100, 400)
            if u < 0.5:   # within the read's true span
                off = qs if fwd else rd_len - qe
                r0 = st + max(0, off) + rnd.randint(-30, 30)
            else:        # far away or another contig
                if rnd.random() < 0.5:
                    q_rf, q_len = rnd.choice(refs)
                r0 = rnd.randint(1, q_len - 1000)
            r0 = max(1, r0)
            qry.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t255\t%s" % (name, rd_len, qs, qe, "+" if q_fwd else "-", q_rf, q_len, r0, r0 + (qe - qs),
                                                                              rnd.randint(20, 60), qe - qs + 1, tags))
        else:
            qry.append("%s\t%d\t*\t*\t*\t*\t*\t*\t*\t*\t*\t255\t%s" % (name, rd_len, tags))
    return "\n".join(qry) + "\n", "# truth\n" + "\n".join(truth) + "\n"


def main():
    spec = importlib.util.spec_from_file_location("ref_pafstats", "/root/reference/uncalled/pafstats.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    OUT.mkdir(exist_ok=True)
    q, t = make_pafs()
    (OUT / "query.paf").write_text(q)
    (OUT / "truth.paf").write_text(t)
    for tag, argv in (("plain", []), ("first20", ["-n", "20"]), ("ref", ["-r", str(OUT / "truth.paf")]),
                      ("annotate", ["-r", str(OUT / "truth.paf"), "-a"])):
        ap = argparse.ArgumentParser()
        ref.add_opts(ap)
        args = ap.parse_args([str(OUT / "query.paf")] + argv)
        so, se = io.StringIO(), io.StringIO()
        with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
            ref.sys.stdout, ref.sys.stderr = so, se
            ref.run(args)
        (OUT / ("%s.stdout" % tag)).write_text(so.getvalue())
        (OUT / ("%s.stderr" % tag)).write_text(se.getvalue())
        print(tag, len(so.getvalue()), len(se.getvalue()))


if __name__ == "__main__":
    main()
