"""Ad-hoc: aggregate a rocprofv3 PC-sampling CSV into a per-PC histogram (the raw file is too large to bring back)."""
import csv, sys, os, collections, glob
d = sys.argv[1]
out = sys.argv[2]
files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)]
with open(out, "w") as o:
    for f in files:
        o.write("== %s  (%d bytes)\n" % (f, os.path.getsize(f)))
        with open(f, newline="") as fh:
            rd = csv.reader(fh)
            try:
                hdr = next(rd)
            except StopIteration:
                continue
            o.write("header: %s\n" % hdr)
            if "pc_sampl" not in os.path.basename(f):
                for i, row in enumerate(rd):
                    if i < 40: o.write("  %s\n" % row)
                continue
            keep = [i for i, h in enumerate(hdr) if any(k in h.lower() for k in ("offset", "instruction", "stall", "reason", "inst_type", "issued", "code_object", "comment"))]
            cnt = collections.Counter()
            n = 0
            for row in rd:
                if n < 5: o.write("  sample: %s\n" % row)
                n += 1
                cnt[tuple(row[i] for i in keep)] += 1
            o.write("samples: %d, distinct: %d; columns %s\n" % (n, len(cnt), [hdr[i] for i in keep]))
            for k, v in cnt.most_common(6000):
                o.write("%d\t%s\n" % (v, "\t".join(k)))
