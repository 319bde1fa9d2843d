"""rocprofv3 --pmc <SQ counters> counter CSV -> per-kernel averages of every collected counter (stdout)."""
import collections, csv, re, sys
import numpy as np

f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "attn_varlen|gemm_kernel"
g = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+<[^>]*>|\w+)\(", name)
    short = m.group(1) if m else name[:40]
    if not re.search(pat, short):
        continue
    g[(short, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), cs in sorted(g.items()):
    n = len(next(iter(cs.values())))
    print(f"{k} grid={grid} launches={n}")
    wc = np.mean(cs["SQ_WAVE_CYCLES"]) if "SQ_WAVE_CYCLES" in cs else None
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
        # SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles: / 32 = the launch's cycles; x 1024 SIMDs = SIMD-cycles
        print(f"    MFMA pipe busy, fraction of SIMD-cycles  {np.mean(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / (np.mean(cs['SQ_BUSY_CYCLES']) * 32):8.3f}"
              f"   (launch = {np.mean(cs['SQ_BUSY_CYCLES']) / 32 / 1e6:.3f} M cycles)")
    for c, v in sorted(cs.items()):
        extra = f"  ({100 * np.mean(v) / wc:5.1f}% of SQ_WAVE_CYCLES)" if wc and c != "SQ_WAVE_CYCLES" else ""
        print(f"    {c:32s} {np.mean(v):16.0f}{extra}")
