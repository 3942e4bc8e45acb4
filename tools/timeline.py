"""One step of a rocprofv3 kernel trace as a text timeline: start / end / duration (us, from the step's first kernel), HW queue, kernel.

usage: python tools/timeline.py <rocprofv3 output dir> [step index, default -2 = the last complete step]
  (cd /tmp; rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 $REPO/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 6 --warmup 4 --pipeline 1)
The step boundary is the stem kernel; with --pipeline 2 the listing interleaves the two lanes (queue ids tell them apart).
"""
import csv, glob, re, sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", n)
    t = (m.group(2) or "").replace("unsigned short", "u16").replace(" ", "")
    return (m.group(1).split("::")[-1] + t)[:48]


def main():
    f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in csv.DictReader(open(f)))
    stems = [i for i, k in enumerate(ks) if k[2].startswith("stem_pool")]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    i0 = stems[which]
    i1 = stems[which + 1] if which != -1 and which + 1 < len(stems) else len(ks)
    t0 = ks[i0][0]
    print("# step of %.1f us (stem to stem), %d launches" % ((ks[i1][0] - t0) / 1e3 if i1 < len(ks) else (ks[-1][1] - t0) / 1e3, i1 - i0))
    print("#  start      end      dur  queue kernel")
    for s, e, n, q in ks[i0:i1]:
        print("%8.1f %8.1f %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))


if __name__ == "__main__":
    main()
