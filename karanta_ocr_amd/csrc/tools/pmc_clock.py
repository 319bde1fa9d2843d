"""rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES counter CSV -> per kernel: launch duration, the clock the chip held
(GRBM_GUI_ACTIVE / 8 XCDs / duration; MI355X_MICROARCH.md 'DVFS give-back'), MFMA pipe busy.
    python karanta_ocr_amd/csrc/tools/pmc_clock.py <counter_collection.csv> [kernel regex]"""
import collections, csv, re, sys
import numpy as np

f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "attn_varlen"
g = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+<[^>]*>|\w+)\(", name)
    short = m.group(1) if m else name[:40]
    if not re.search(pat, short):
        continue
    key = (short, r["Grid_Size"])
    g[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        g[key]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for (k, grid), cs in sorted(g.items()):
    ns = np.array(cs["_ns"]); gui = np.array(cs["GRBM_GUI_ACTIVE"])
    line = f"{k} grid={grid} launches={len(ns)}  {np.median(ns) / 1e3:9.1f} us  clock {np.median(gui / 8 / ns):5.2f} GHz"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
        line += f"  MFMA busy {np.mean(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / (np.mean(cs['SQ_BUSY_CYCLES']) * 32):.3f}"
    print(line)
