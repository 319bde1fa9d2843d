"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) average duration of the decode kernels."""
import collections, csv, glob, re, sys
import numpy as np

f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof*/*/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
g = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"(\w+<[^>]*>|\w+)\(", n.replace("(anonymous namespace)::", ""))
    key = ((m.group(1) if m else n)[-52:], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"])
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in g.values())
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:54s} grid {k[1]:>8s}x{k[2]:<2s} wg {k[3]:>4s} vgpr {k[4]:>3s} lds {k[5]:>6s} n={len(v):6d} avg {np.mean(v):9.2f} us  min {np.min(v):9.2f}  share {100*sum(v)/tot:5.1f}%")
