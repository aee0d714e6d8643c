"""Compact a rocprofv3 *_kernel_stats.csv into profiles/<name>.csv (kernel names shortened)."""
import csv, glob, re, sys
src_dir, out, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
f = (glob.glob(f"{src_dir}/*/*kernel_stats.csv") + glob.glob(f"{src_dir}/*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
lines = [f"# {note}", "Name,Calls,TotalDurationNs,AverageNs,Percentage"]
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = n if len(n) <= 110 else n[:107] + "..."
    lines.append(f'"{n}",{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.0f},{r["Percentage"]}')
open(out, "w").write("\n".join(lines) + "\n")
for l in lines[:24]:
    print(l[:150])
