"""How busy was the GPU?  Union of the kernel intervals of a rocprofv3 --kernel-trace CSV against the span they cover, and
where the idle time sits: gaps by size class, and for the long gaps which kernel ended before / started after them.

    python gpu_busy.py KERNEL_TRACE.csv [--skip-first-s S | --last-s S] [--long-us 200]

Used on the corpus run (bench_corpus under rocprofv3): pages/s there is set by the scheduler thread keeping the one GPU
stream fed, so the number that matters is busy / span, not any single kernel."""
import argparse, collections, csv, re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--skip-first-s", type=float, default=0.0, help="drop this much of the trace's start (load, warm-up, capture)")
ap.add_argument("--last-s", type=float, default=0.0, help="keep only this much of the trace's end (the timed region of a run)")
ap.add_argument("--long-us", type=float, default=200.0)
a = ap.parse_args()


def short(n):
    m = re.search(r"(\w+<[^>]*>|\w+)\(", n.replace("(anonymous namespace)::", ""))
    return (m.group(1) if m else n)[-44:]


iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(a.csv)))
t_begin = iv[0][0] + int(a.skip_first_s * 1e9)
if a.last_s > 0:
    t_begin = max(t_begin, max(e for _, e, _ in iv) - int(a.last_s * 1e9))
iv = [x for x in iv if x[0] >= t_begin]
span = (max(e for _, e, _ in iv) - iv[0][0]) / 1e3
busy, cur_e, prev = 0.0, iv[0][0], None
gaps = []
for s, e, n in iv:
    if s > cur_e:
        gaps.append(((s - cur_e) / 1e3, prev, n))
        busy += (e - s) / 1e3
        cur_e, prev = e, n
    elif e > cur_e:
        busy += (e - cur_e) / 1e3
        cur_e, prev = e, n
print(f"kernels {len(iv)}  span {span / 1e6:.3f} s  busy {busy / 1e6:.3f} s  = {100 * busy / span:.1f} %  idle {(span - busy) / 1e6:.3f} s")
classes = [(0, 3), (3, 10), (10, 50), (50, 200), (200, 1000), (1000, 10000), (10000, 1e12)]
for lo, hi in classes:
    g = [x[0] for x in gaps if lo <= x[0] < hi]
    print(f"  gaps {lo:>6} .. {hi:<8g} us: n={len(g):7d}  total {sum(g) / 1e3:10.2f} ms  ({100 * sum(g) / span:5.2f} % of the span)")
by = collections.defaultdict(lambda: [0, 0.0])
for d, p, n in gaps:
    if d >= a.long_us:
        by[(p, n)][0] += 1
        by[(p, n)][1] += d
print(f"gaps >= {a.long_us:g} us by (kernel before -> kernel after):")
for (p, n), (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {d / 1e3:9.2f} ms  n={c:5d}  avg {d / c:8.1f} us   {p}  ->  {n}")
