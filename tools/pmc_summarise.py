"""Folds the rocprofv3 passes of tools/profile_round.sh into two summaries:
  <out>/kernel_stats.csv   per-kernel calls / total / average / min / max (ns) from --stats
  <out>/pmc_traffic.json   per-kernel FETCH_SIZE / WRITE_SIZE means, corrected per
                           /opt/skills/guides/MI355X_MICROARCH.md (counter unit KiB; gfx950 reports half of
                           streamed reads: checked against the 1 GiB calibration kernels), and the
                           bytes-per-launch figures bench.py prints as roofline.traffic."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pattern):
    hits = sorted(glob.glob(os.path.join(out, sub, "**", pattern), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = name.split("(")[0].strip()
    return name[:-3] if name.endswith(".kd") else name


def counters(sub, counter):
    path = find(sub, "*counter_collection.csv")
    per = defaultdict(list)
    if not path:
        return per
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == counter:
                per[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return per


def summarise(per):
    return {k: {"dispatches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)}
            for k, v in per.items()}


raw = {}
for prefix, fsub, wsub in (("calib", "calib_fetch", "calib_write"), ("bench", "fetch", "write")):
    f = summarise(counters(fsub, "FETCH_SIZE"))
    w = summarise(counters(wsub, "WRITE_SIZE"))
    for k in sorted(set(f) | set(w)):
        raw[f"{prefix}:{k}"] = {"FETCH_SIZE": f.get(k), "WRITE_SIZE": w.get(k)}

# unit check on the calibration kernels: each reads or writes 1 GiB = 1048576 KiB
fetch_scale = 2.0
cal = raw.get("calib:calib_read4", {}).get("FETCH_SIZE")
if cal:
    fetch_scale = 1048576.0 / cal["mean_KiB"]


def traffic(kernel_prefix):
    ks = [k for k in raw if k.startswith("bench:") and kernel_prefix in k]
    if not ks:
        return None
    fe = sum(raw[k]["FETCH_SIZE"]["mean_KiB"] * raw[k]["FETCH_SIZE"]["dispatches"] for k in ks if raw[k]["FETCH_SIZE"])
    nd = sum(raw[k]["FETCH_SIZE"]["dispatches"] for k in ks if raw[k]["FETCH_SIZE"])
    wr = sum(raw[k]["WRITE_SIZE"]["mean_KiB"] * raw[k]["WRITE_SIZE"]["dispatches"] for k in ks if raw[k]["WRITE_SIZE"])
    nw = sum(raw[k]["WRITE_SIZE"]["dispatches"] for k in ks if raw[k]["WRITE_SIZE"])
    if not nd or not nw:
        return None
    return {"read_bytes": int(fe / nd * 1024 * fetch_scale), "write_bytes": int(wr / nw * 1024)}


it = traffic("k_iterate<")
ba = traffic("BatchCfg<1024")          # the 1024-thread variant does the config-4 pairs; the 256-thread pre-pass only marks them
i3 = traffic("k_iterate3")
b3 = traffic("k_batch3<")              # the 3D loop-closure batch (256 config-5-sized pairs per launch in bench.py)
summary = {
    "source": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes "
              "(--kernel-trace only) of `bench.py --no-cpu-baseline --headline-only --steps 10 --warmup 2` on one MI355X",
    "units": "counter values are KiB; FETCH_SIZE scaled by the factor measured on the 1 GiB calibration reads "
             f"(tools/pmc_calib.hip): {fetch_scale:.4f} (MI355X_MICROARCH.md: gfx950 reports half of streamed "
             "reads); WRITE_SIZE taken as is (calibration: 1 GiB of stores reads 1048576 KiB)",
    "fetch_scale": fetch_scale,
    "raw": raw,
}
if it:
    summary["k_iterate"] = dict(it, algorithmic_bytes=3200000)
    summary["bytes_per_launch"] = it["read_bytes"] + it["write_bytes"]
if i3:
    summary["k_iterate3"] = dict(i3, algorithmic_bytes=6815744)
    summary["bytes_per_launch_3d"] = i3["read_bytes"] + i3["write_bytes"]
if ba:
    summary["k_batch"] = dict(ba, pairs=512)
    summary["batch_bytes_per_launch"] = ba["read_bytes"] + ba["write_bytes"]
if b3:
    summary["k_batch3"] = dict(b3, pairs=256, algorithmic_bytes=256 * 131072 * 12 * 31)
    summary["batch3_bytes_per_launch"] = b3["read_bytes"] + b3["write_bytes"]
# the grid build (round 3: k_bounds_parts -> k_chunk_sort -> k_tile_gather).  The profiled command builds the 1M-point
# config-3 submap (ndt2d_set_target_dev) and merges the 100k-point scan into it (ndt2d_add_target_points_dev) the same
# number of times, so k_chunk_sort / k_tile_gather averages mix the two; the per-call split is in the kernel trace.
gb = {}
for name in ("k_bounds_parts", "k_chunk_sort<16>", "k_chunk_sort<4>", "k_tile_gather"):
    t = traffic(name)
    if t:
        gb[name] = t
if gb:
    summary["grid_build_kernels"] = gb
with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
    json.dump(summary, f, indent=1)

# SQ pass: per kernel, the mean of each counter over its dispatches and two ratios the design notes quote
sq_path = find("sq", "*counter_collection.csv")
if sq_path:
    per = defaultdict(lambda: defaultdict(list))
    with open(sq_path, newline="") as f:
        for row in csv.DictReader(f):
            per[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    sq = {}
    for k, cs in per.items():
        if not any(t in k for t in ("k_iterate", "k_batch", "k_align_small", "k_tile", "k_chunk_sort", "k_bounds_parts")):
            continue
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["dispatches"] = len(next(iter(cs.values())))
        if e.get("SQ_WAVE_CYCLES"):
            e["valu_active_share_of_wave_cycles"] = e.get("SQ_ACTIVE_INST_VALU", 0.0) / e["SQ_WAVE_CYCLES"]
            e["wait_any_share_of_wave_cycles"] = e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share_of_lds_cycles"] = e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"]
        if e.get("SQ_BUSY_CYCLES"):
            e["lds_active_share_of_busy_cycles"] = e.get("SQ_LDS_IDX_ACTIVE", 0.0) / e["SQ_BUSY_CYCLES"]
        sq[k] = e
    with open(os.path.join(out, "sq_summary.json"), "w") as f:
        json.dump({"source": "tools/profile_round.sh pass 5 (rocprofv3 --pmc SQ_*, --kernel-trace only); SQ_WAVE_CYCLES / SQ_WAIT_* / "
                             "SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)", "kernels": sq}, f, indent=1)

mk = find("markers", "*marker_api_stats.csv") or find("markers", "*marker*stats.csv")
if mk:
    with open(mk) as f, open(os.path.join(out, "marker_stats.csv"), "w") as g:
        g.write(f.read())

stats = find("stats", "*kernel_stats.csv")
if stats:
    with open(stats) as f, open(os.path.join(out, "kernel_stats.csv"), "w") as g:
        g.write(f.read())
print("bytes_per_launch", summary.get("bytes_per_launch"), "batch_bytes_per_launch", summary.get("batch_bytes_per_launch"),
      "fetch_scale", round(fetch_scale, 4))
