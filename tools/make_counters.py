#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc_summary.txt (tools/pmc_collect.sh) -> profiles/counters.json: per path stage the HBM bytes per
launch (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction), VALU wave-instructions, VALU-busy fraction and LDS
bank-conflict cycles, stamped with the hash of the kernel sources they were measured on (bench.csrc_sha): bench.py
prints them only while that hash matches the sources it runs.  Also copies the summaries into profiles/.
usage: python tools/make_counters.py <tag>      (run in the build container after the gpurun call)"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
out = os.path.join(ROOT, "gpurun_out")
vals = {}
for line in open(os.path.join(out, f"{tag}_pmc_summary.txt")):
    if line.startswith("=="):
        continue
    m = re.match(r"(\S.*?)\s{2,}(\S+=.*)$", line.rstrip())
    if not m:
        continue
    d = vals.setdefault(m.group(1).strip(), {})
    for kv in m.group(2).split():
        k, v = kv.split("=")
        d[k] = float(v)
# kernel name prefix -> path stage (stages with several kernels: per-launch values are added up over one frame's launches)
STAGE = {"render_bwd": ["render_bwd_kernel"], "render_fwd": ["render_fwd_kernel"], "preprocess_fwd": ["preprocess_fwd_kernel"],
         "preprocess_bwd": ["preprocess_bwd_kernel"], "duplicate_with_keys": ["duplicate_with_keys_kernel"],
         "identify_tile_ranges": ["identify_tile_ranges_kernel"],
         "scan_tiles": ["scan_reduce_rect_kernel", "scan_sums_kernel", "scan_final_rect_kernel", "scan_reduce_kernel", "scan_final_kernel"],
         "radix_sort_pass": ["radix_hist_kernel", "radix_scatter_kernel"]}
# mean issue cost of the visit loops' instruction mix (profiles/r1f_valu_issue_rates.txt; counted in the ISA)
ISSUE = {"render_bwd": float(os.environ.get("GS_BWD_ISSUE_CYCLES", "3.64")), "render_fwd": 3.57}
# Several instantiations of one kernel can appear in a trace (the bench also renders once without a backward to follow:
# preprocess_fwd_kernel<3, true, false>).  A stage counts each kernel name once: the instantiation the timed path
# launches, i.e. the one with the most calls in the kernel trace.
calls = {}
stats_csv = os.path.join(out, f"{tag}_kernel_stats.csv")
if os.path.exists(stats_csv):
    for r in csv.DictReader(open(stats_csv)):
        m = re.search(r"::(\w+_kernel)(<.*?>)?\(", r["Name"])
        if m:
            calls[m.group(1) + (m.group(2) or "")] = int(r["Calls"])
best = {}
for name in vals:
    base = name.split("<")[0]
    if base not in best or calls.get(name, 0) > calls.get(best[base], 0):
        best[base] = name
kernels = {}
for stage, prefixes in STAGE.items():
    acc = {}
    for name, d in vals.items():
        if best.get(name.split("<")[0]) != name:
            continue
        if any(name.startswith(p) for p in prefixes):
            for k, v in d.items():
                acc[k] = acc.get(k, 0.0) + v
    if not acc:
        continue
    e = {}
    if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:      # rocprofv3 reports KB
        e["fetch_size_raw_bytes"] = int(acc["FETCH_SIZE"] * 1024)
        e["write_size_bytes"] = int(acc["WRITE_SIZE"] * 1024)
        e["hbm_traffic_bytes"] = 2 * e["fetch_size_raw_bytes"] + e["write_size_bytes"]
    if "SQ_INSTS_VALU" in acc:
        e["valu_wave_insts"] = acc["SQ_INSTS_VALU"]
    if "SQ_ACTIVE_INST_VALU" in acc and acc.get("SQ_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES is summed over the 32 shader engines' SQs ... calibrated as in profiles/README.md:
        # busy fraction = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES x 32)
        e["valu_busy_frac"] = round(acc["SQ_ACTIVE_INST_VALU"] * 4 / (acc["SQ_BUSY_CYCLES"] * 32), 4)
    if "SQ_LDS_BANK_CONFLICT" in acc:
        e["lds_bank_conflict_cycles"] = acc["SQ_LDS_BANK_CONFLICT"]
        e["lds_active_cycles"] = acc.get("SQ_LDS_IDX_ACTIVE")
    if stage in ISSUE:
        e["mean_issue_cycles"] = ISSUE[stage]
    kernels[stage] = e
b = json.load(open(os.path.join(out, f"{tag}_bench.json")))
res = {"source": f"profiles/{tag}_pmc_summary.txt (tools/pmc_collect.sh: one rocprofv3 --pmc run per counter group on "
                 "`bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-mode --no-next-rows`; means per launch)",
       "correction": "hbm_traffic_bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts "
                     "128-byte requests as 64)",
       "csrc_sha": bench.csrc_sha(), "config": b["config"]["workload"], "t_eps": b["config"]["t_eps"], "kernels": kernels}
json.dump(res, open(os.path.join(ROOT, "profiles", "counters.json"), "w"), indent=1)
for src, dst in ((f"{tag}_pmc_summary.txt", f"{tag}_pmc_summary.txt"), (f"{tag}_kernel_stats.csv", f"{tag}_kernel_stats_config3.csv"),
                 (f"{tag}_bench.json", f"{tag}_bench_config3.json"), (f"{tag}_bench_under_rocprof.json", f"{tag}_bench_config3_under_rocprof.json")):
    if os.path.exists(os.path.join(out, src)):
        shutil.copy(os.path.join(out, src), os.path.join(ROOT, "profiles", dst))
print(json.dumps(res, indent=1)[:1500])
