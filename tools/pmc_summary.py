#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory (gpurun_out/prof_<tag>/) into the
small, committed files under profiles/:

    profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
    profiles/<round>_<tag>_pmc.json           per-launch means of every counter collected
    profiles/pmc_traffic.json                 {workload: {"hbm_bytes_per_launch": ...}} read by bench.py

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB and
come from separate --pmc passes; on gfx950 FETCH_SIZE reports half the bytes of a coalesced
streaming read, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  The
decode kernels read with 4-byte-per-lane loads, a width the guide calls uncalibrated, so
the doubled figure is cross-checked against the algorithmic read bytes and both are kept.

    python tools/pmc_summary.py gpurun_out/prof_4k r01 4k
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed_dispatches(kernel_trace_csv, kernel_filter):
    """Durations (ns) of the `kernel_filter` dispatches between an opening and a closing sentinel of bench.py, by kernel
    name; None when the trace holds no sentinel pair.  Sentinels: kernel copy_probe with 512 (opening) or 1024 (closing)
    work-items in the launch."""
    rows = list(csv.DictReader(open(kernel_trace_csv)))
    if not rows or "Grid_Size" not in rows[0] and "Grid_Size_X" not in rows[0]:
        return None

    def grid(r):
        if "Grid_Size" in r:
            return int(r["Grid_Size"])
        return int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by_kernel, regions, inside, pending = collections.defaultdict(list), 0, False, []
    for r in rows:
        name = r["Kernel_Name"]
        if "copy_probe" in name and grid(r) in (512, 1024):
            if grid(r) == 512:
                inside, pending = True, []  # a region that never closed is dropped
            elif inside:
                inside = False
                for k, d in pending:  # committed only when the closing sentinel has been seen
                    by_kernel[k].append(d)
                regions += 1 if pending else 0
                pending = []
            continue
        if inside and kernel_filter in name:
            pending.append((name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return {"regions": regions, "by_kernel": dict(by_kernel)} if regions else None


def timed_dispatch_ids(pmc_rows):
    """Dispatch ids of a --pmc pass that lie between bench.py's sentinels (one stream: dispatch order = stream order), or None
    when the pass holds no sentinel pair.  The set-up's placement probes run the same kernel over UNFILLED (constant) memory,
    which the LDS gathers of the rescale kernels serve without bank conflicts: averaged in, they make every counter (and
    round 3's rocprof mean, 384 us against 413 us timed) look better than the timed launches are."""
    by_id = {}
    for r in pmc_rows:
        by_id.setdefault(int(r["Dispatch_Id"]), r)
    keep, inside, pairs, pending = set(), False, 0, []
    for did in sorted(by_id):
        r = by_id[did]
        if "copy_probe" in r["Kernel_Name"] and int(r["Grid_Size"]) in (512, 1024):
            if int(r["Grid_Size"]) == 512:
                inside, pending = True, []
            elif inside:
                inside, pairs = False, pairs + 1
                keep.update(pending)
                pending = []
            continue
        if inside:
            pending.append(r["Dispatch_Id"])
    return keep if pairs else None


def write_timed_stats(path, timed):
    total = sum(sum(v) for v in timed["by_kernel"].values())
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "MedianNs", "Scope"])
        for name, v in sorted(timed["by_kernel"].items(), key=lambda kv: -sum(kv[1])):
            mean = sum(v) / len(v)
            sd = (sum((x - mean) ** 2 for x in v) / len(v)) ** 0.5
            w.writerow([name, len(v), sum(v), "%.3f" % mean, "%.2f" % (100.0 * sum(v) / total), min(v), max(v), "%.3f" % sd,
                        sorted(v)[len(v) // 2],
                        "dispatches inside bench.py's %d sentinel-bracketed timed regions only" % timed["regions"]])


def main():
    src, rnd, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    kernel_filter = sys.argv[4] if len(sys.argv) > 4 else "decode_nv12"
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    stats = os.path.join(src, "trace", "trace_kernel_stats.csv")
    kt = os.path.join(src, "trace", "trace_kernel_trace.csv")
    timed = timed_dispatches(kt, kernel_filter) if os.path.exists(kt) else None
    if timed:
        # bench.py brackets every region that counts towards `value` with two sentinel dispatches (copy_probe, 512 lanes
        # opening / 1024 closing, outside both clocks).  <round>_<tag>_kernel_stats.csv = the rocprofv3 stats columns computed
        # over the dispatches BETWEEN sentinels only (the set-up's placement probes, warm-up, spot check and side legs run the
        # same kernel and would otherwise be averaged in: round 3's mean was 2 % above ms_per_step for that reason);
        # rocprofv3's own all-dispatch summary is kept beside it as ..._kernel_stats_all.csv.
        if os.path.exists(stats):
            shutil.copy(stats, os.path.join(out_dir, "%s_%s_kernel_stats_all.csv" % (rnd, tag)))
        write_timed_stats(os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (rnd, tag)), timed)
    elif os.path.exists(stats):
        shutil.copy(stats, os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (rnd, tag)))

    # durations from the kernel trace of the --stats run
    durs = []
    if timed:
        durs = [d for v in timed["by_kernel"].values() for d in v]
    elif os.path.exists(kt):
        for r in csv.DictReader(open(kt)):
            if kernel_filter in r["Kernel_Name"]:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    counters = collections.defaultdict(list)
    meta, scope = {}, {}
    for d in sorted(os.listdir(src)):
        f = os.path.join(src, d, "pmc_counter_collection.csv")
        if not (d.startswith("pmc_") and os.path.exists(f)):
            continue
        rows = list(csv.DictReader(open(f)))
        keep = timed_dispatch_ids(rows)  # None: no sentinels in this pass (another program than bench.py): every dispatch counts
        for r in rows:
            if kernel_filter in r["Kernel_Name"] and (keep is None or r["Dispatch_Id"] in keep):
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                          "VGPR_Count", "SGPR_Count")}
        scope["pmc_" + d[4:]] = "timed regions only" if keep is not None else "all dispatches"
    mean = {k: sum(v) / len(v) for k, v in counters.items()}
    summary = {"source": os.path.relpath(src, ROOT), "kernel": meta, "dispatch_scope": scope, "launches_profiled": {k: len(v) for k, v in counters.items()},
               "per_launch_mean": mean}
    if durs:
        durs.sort()
        summary["duration_ns"] = {"n": len(durs), "mean": sum(durs) / len(durs), "median": durs[len(durs) // 2],
                                  "min": durs[0], "max": durs[-1]}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        fetch_raw = mean["FETCH_SIZE"] * 1024
        write = mean["WRITE_SIZE"] * 1024
        summary["hbm"] = {
            "fetch_bytes_raw": fetch_raw,
            "fetch_bytes_gfx950_corrected": 2 * fetch_raw,
            "write_bytes": write,
            "hbm_bytes_per_launch": 2 * fetch_raw + write,
            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); "
                    "WRITE_SIZE exact for 16 B/lane stores",
        }
    if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean:
        summary["l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
    if "GRBM_GUI_ACTIVE" in mean and durs:
        summary["effective_clock_ghz"] = mean["GRBM_GUI_ACTIVE"] / 8 / (summary["duration_ns"]["mean"])
    if "SQ_LDS_IDX_ACTIVE" in mean and "SQ_INSTS_LDS" in mean:
        summary["lds_cycles_per_instruction"] = mean["SQ_LDS_IDX_ACTIVE"] / mean["SQ_INSTS_LDS"]
    json.dump(summary, open(os.path.join(out_dir, "%s_%s_pmc.json" % (rnd, tag)), "w"), indent=1)

    if "hbm" in summary:
        tpath = os.path.join(out_dir, "pmc_traffic.json")
        traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
        traffic[tag] = {"hbm_bytes_per_launch": summary["hbm"]["hbm_bytes_per_launch"], "round": rnd,
                        "fetch_bytes_corrected": summary["hbm"]["fetch_bytes_gfx950_corrected"],
                        "write_bytes": summary["hbm"]["write_bytes"]}
        json.dump(traffic, open(tpath, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
