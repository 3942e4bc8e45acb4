#!/usr/bin/env python3
"""GPU busy-time analysis of a rocprofv3 --kernel-trace CSV: union of kernel intervals vs wall, top kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# middle of the trace (steady state of a `--no-roofline` run with many steps)
span = iv[-1][1] - iv[0][0]
t_lo, t_hi = iv[0][0] + 0.5 * span, iv[0][0] + 0.95 * span
iv = [x for x in iv if x[0] >= t_lo and x[1] <= t_hi]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = iv[-1][1] - iv[0][0]
tot = sum(e - s for s, e, _ in iv)
print("window %.2f ms: busy(union) %.2f ms = %.1f%%, sum of kernel durations %.2f ms (overlap factor %.2f)" % (wall / 1e6, busy / 1e6, 100.0 * busy / wall, tot / 1e6, tot / busy))
agg = collections.Counter()
for s, e, n in iv: agg[n[:70]] += e - s
for n, t in agg.most_common(12): print("  %6.2f%%  %s" % (100.0 * t / tot, n))
