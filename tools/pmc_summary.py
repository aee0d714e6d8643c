"""Summarise rocprofv3 --pmc passes (one counter_collection.csv per pass, any number of passes of the SAME command) per
libicl_hip kernel and launch geometry (grid size = the shape class of a GEMM / attention launch):
launches, mean wall duration, mean of every counter, and the derived figures of MI355X_MICROARCH.md:
  clock_ghz      = GRBM_GUI_ACTIVE / 8 / duration     (rocprofv3 sums the counter over the 8 XCDs)
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)   (busy cycles summed over SIMDs)
usage: pmc_summary.py <dir-with-pass-subdirs> <out.json> [note]"""
import csv, glob, json, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
root, out_path, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
files = sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True))
groups = defaultdict(lambda: {"n": defaultdict(int), "sum": defaultdict(float), "dur": 0.0, "dur_n": 0})
for f in files:
    seen = set()
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if "at::native" in name or name.startswith("__amd"):
            continue
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        key = (name, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        g = groups[key]
        g["sum"][r["Counter_Name"]] += float(r["Counter_Value"])
        g["n"][r["Counter_Name"]] += 1
        did = (f, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did)
            g["dur"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            g["dur_n"] += 1
            g["vgpr"] = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
            g["lds"] = int(r["LDS_Block_Size"])
rows = []
for (name, grid, wg), g in groups.items():
    c = {k: g["sum"][k] / g["n"][k] for k in g["sum"]}
    dur = g["dur"] / max(g["dur_n"], 1)
    row = {"kernel": name, "grid_threads": grid, "workgroups": grid // wg, "workgroup_size": wg, "vgprs": g.get("vgpr"),
           "lds_bytes": g.get("lds"), "launches_per_pass": g["dur_n"] // max(len(files), 1), "mean_duration_us": round(dur / 1e3, 2),
           "counters": {k: round(v, 1) for k, v in sorted(c.items())}}
    gui = c.get("GRBM_GUI_ACTIVE")
    if gui and dur > 0:
        row["clock_ghz"] = round(gui / 8.0 / dur, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0), 4)
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if k in c:
                row[k.lower() + "_per_wave_cycle"] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
    row["total_us_per_pass"] = round(dur * row["launches_per_pass"] / 1e3, 1)
    rows.append(row)
rows.sort(key=lambda r: -r["total_us_per_pass"])
json.dump({"note": note, "passes": len(files), "derived": "clock_ghz = GRBM_GUI_ACTIVE/8/duration; mfma_busy_frac = "
           "SQ_VALU_MFMA_BUSY_CYCLES/(GRBM_GUI_ACTIVE/8*1024 SIMDs); durations are those of the PROFILED passes", "kernels": rows},
          open(out_path, "w"), indent=1)
for r in rows[:24]:
    print(f"{r['kernel'][:44]:44s} wgs {r['workgroups']:6d} x{r['launches_per_pass']:5d} {r['mean_duration_us']:9.1f} us  clk {r.get('clock_ghz', 0):5.3f}  "
          f"mfma {r.get('mfma_busy_frac', 0):6.4f}")
