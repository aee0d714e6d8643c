"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into the
per-launch HBM traffic of each libicl_hip kernel, with the gfx950 correction of MI355X_MICROARCH.md §HBM:
FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-B-per-lane stores -> write bytes = WRITE_SIZE * 1024 (both counters are in KiB)."""
import csv, glob, json, re, sys
from collections import defaultdict

import os
# "kernel name:grid,grid,...=label": launches of that kernel with one of those Grid_Size values are listed under their own key
# (round 4: the decode GEMMs now run on gemm256_bf16_kernel<false, 16> with split-K — 128 / 172 / 192 workgroups — inside the
# captured decode graph; the roofline of the dominant kernel is taken over the launches the live HIP events time, i.e. without them)
SPLIT = {}
for spec in filter(None, os.environ.get("ICL_PMC_SPLIT", "").split(";")):
    key, label = spec.rsplit("=", 1)
    name, grids = key.rsplit(":", 1)
    SPLIT[name] = ({int(g) for g in grids.split(",")}, label)

def collect(d, counter):
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if n in SPLIT and int(r["Grid_Size"]) in SPLIT[n][0]:
            n = f"{n} [{SPLIT[n][1]}]"
        acc[n][0] += float(r["Counter_Value"]); acc[n][1] += 1
    return acc

fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if not (k.startswith("gemm") or k.startswith("attn") or k.startswith("norm") or "kernel" in k and "at::" not in k):
        continue
    rd = 2.0 * fetch[k][0] * 1024 / fetch[k][1]
    wr = write[k][0] * 1024 / write[k][1] if k in write else 0.0
    out[k] = {"launches": fetch[k][1], "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
              "hbm_bytes_per_launch": round(rd + wr)}
json.dump({"command": sys.argv[3] if len(sys.argv) > 3 else "", "correction": "read = 2*FETCH_SIZE KiB (gfx950), write = WRITE_SIZE KiB",
           "kernels": out}, open(sys.argv[4] if len(sys.argv) > 4 else "/dev/stdout", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(f"{k[:60]:60s} launches {v['launches']:5d}  read {v['read_bytes_per_launch']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:9.1f} MB")
