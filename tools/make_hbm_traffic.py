#!/usr/bin/env python3
"""profiles/<round>_pmc_hbm_counters.txt (tools/pmc_summary.py output, FETCH_SIZE / WRITE_SIZE in KB per launch)
-> profiles/hbm_traffic.json: HBM bytes per launch and kernel with the guide's gfx950 correction
(traffic = 2 x FETCH_SIZE + WRITE_SIZE).  Template instantiations of one kernel are added up when they belong to one
stage of a frame (preprocess_bwd = geometry + SH launches), otherwise the largest is kept.
usage: make_hbm_traffic.py profiles/r1e_pmc_hbm_counters.txt > profiles/hbm_traffic.json"""
import json, re, sys
src = sys.argv[1]
vals = {}
for line in open(src):
    m = re.match(r"(\S.*?)\s+((?:FETCH_SIZE|WRITE_SIZE)=\S+)\s*$", line)
    if not m:
        continue
    name, (key, v) = m.group(1).strip(), m.group(2).split("=")
    vals.setdefault(name, {})[key] = float(v) * 1024.0
out = {}
for name, d in vals.items():
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d or name.startswith("at::"):
        continue
    base = name.split("<")[0]
    e = out.setdefault(base, {"fetch_size_raw_bytes": 0, "write_size_bytes": 0})
    if base == "preprocess_bwd_kernel":          # two launches per frame
        e["fetch_size_raw_bytes"] += int(d["FETCH_SIZE"]); e["write_size_bytes"] += int(d["WRITE_SIZE"])
    elif d["FETCH_SIZE"] + d["WRITE_SIZE"] > e["fetch_size_raw_bytes"] + e["write_size_bytes"]:
        e["fetch_size_raw_bytes"], e["write_size_bytes"] = int(d["FETCH_SIZE"]), int(d["WRITE_SIZE"])
for e in out.values():
    e["traffic_bytes"] = 2 * e["fetch_size_raw_bytes"] + e["write_size_bytes"]
json.dump({"source": f"{src} (tools/pmc_hbm.sh: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE, "
                     "config 3, t_eps 1e-5; means over the launches of one bench.py run, radix kernels over both sorts)",
           "correction": "traffic = 2 x FETCH_SIZE + WRITE_SIZE per launch (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE "
                         "tallies 128-B read requests at 64 B; WRITE_SIZE is exact for 16-B stores and float atomics; "
                         "Infinity-Cache hits are included)",
           "kernels": out}, sys.stdout, indent=1)
