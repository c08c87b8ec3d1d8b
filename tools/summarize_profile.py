"""Turn the raw rocprofv3 outputs under gpurun_out/ (tools/profile_gpu.sh) into the small,
tracked summaries under profiles/.  usage: python tools/summarize_profile.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

shutil.copy(os.path.join(OUT, "prof_stats", "stats_kernel_stats.csv"),
            os.path.join(dst, "%s_kernel_stats.csv" % tag))
bench = open(os.path.join(OUT, "prof_stats_bench.json")).read().strip().splitlines()[-1]
open(os.path.join(dst, "%s_bench_under_rocprof.json" % tag), "w").write(bench + "\n")

rows = list(csv.DictReader(open(os.path.join(OUT, "prof_stats", "stats_kernel_stats.csv"))))
nf = [int(r["Calls"]) for r in rows if "render_forward" in r["Name"]][0]
lines = ["# rocprofv3 --kernel-trace --stats of `python bench.py --steps N --warmup 5 --no-cpu-baseline --no-train --no-strong --no-delivery`",
         "# (%d forward calls: timed frames + warm-up + the untimed V/R statistics pass)" % nf,
         "# per-frame = TotalDurationNs / forward calls", "",
         "%10s %8s %10s  %s" % ("us/frame", "calls/fr", "avg us", "kernel")]
tot = 0.0
for r in rows:
    per = float(r["TotalDurationNs"]) / nf / 1000
    tot += per
    lines.append("%10.1f %8.1f %10.1f  %s" % (per, int(r["Calls"]) / nf, float(r["AverageNs"]) / 1000,
                                             r["Name"].split("(")[0][-70:]))
lines.append("%10.1f  total GPU-busy us per frame" % tot)
# the default command runs priming + warm-up + timed frames over three streams (kernels of several
# frames share the chip: a launch's wall time is not its own speed) and then single-stream passes
# (statistics, stage timing, latency): the render kernel's duration distribution separates the two
try:
    tr = [r for r in csv.DictReader(open(os.path.join(OUT, "prof_stats", "stats_kernel_trace.csv")))
          if "render_forward_kernel<false" in r["Kernel_Name"]]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr)
    q = lambda f: d[min(len(d) - 1, int(f * len(d)))]   # noqa: E731
    solo = [x for x in d if x <= 1.08 * q(0.5)]
    lines += ["", "# render_forward_kernel<false,...> launch durations (us), %d launches: min %.1f  p25 %.1f  median %.1f  "
              "p75 %.1f  p90 %.1f  max %.1f" % (len(d), d[0], q(0.25), q(0.5), q(0.75), q(0.9), d[-1]),
              "# launches with one frame in flight (within 8 %% of the median): %d, mean %.1f us;"
              " the others ran inside the three-stream region" % (len(solo), sum(solo) / len(solo))]
except Exception as exc:
    lines.append("# (no kernel trace: %r)" % (exc,))

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(OUT, "prof_pmc*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("grpg::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines += ["", "# PMC (separate rocprofv3 --pmc passes, mean per dispatch).",
          "# FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide",
          "# coalesced reads by 2x on gfx950 (reported raw here, corrected in DESIGN.md)."]
for k in sorted(agg):
    if not any(x in k for x in ("render_forward", "preprocess_kernel", "radix_scatter", "emit_kernel",
                                "radix_hist", "scan_down", "scan_reduce", "tile_ranges", "depth_scatter",
                                "depth_hist", "frame_init", "pack_u8", "hb_", "radix_digit_scan",
                                "publish_counts", "depth_")):
        continue
    lines.append("== %s" % k)
    for c, v in sorted(agg[k].items()):
        lines.append("   %-24s %16.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
open(os.path.join(dst, "%s_summary.txt" % tag), "w").write("\n".join(lines) + "\n")

# HBM traffic of the dominant kernel for bench.py's roofline.traffic (bytes per launch):
# WRITE_SIZE as reported; FETCH_SIZE x2 (gfx950 correction for 16-byte-per-lane loads,
# MI355X_MICROARCH.md "HBM": calibrated here on preprocess_kernel whose read volume is known).
bj = json.loads(bench)
traffic = {"workload": {k: bj["config"].get(k) for k in ("P", "width", "height", "M")}}
for k in agg:
    if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
        f = sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"]) * 1024.0
        w = sum(agg[k]["WRITE_SIZE"]) / len(agg[k]["WRITE_SIZE"]) * 1024.0
        # template arguments stay in the key (radix_scatter_kernel<7, 8> and <8, 8> are different
        # launches); the render kernel is additionally listed under its bare name for bench.py
        entry = {"fetch_bytes_raw": f, "write_bytes": w, "hbm_bytes_corrected": 2.0 * f + w,
                 "launches_sampled": len(agg[k]["FETCH_SIZE"])}
        # Round 6 calibration (tools/ubench/gather_fetch.hip, profiles/round6_fetch_calibration.json): FETCH_SIZE
        # counts a wide STREAMING read at 1/2 of its bytes (the guide's case) but a GATHERED 64-byte record at 64
        # bytes, and WRITE_SIZE is exact.  So 2 F + W (hbm_bytes_corrected) is right for the streaming kernels
        # (preprocess, the sort and scan passes) and an UPPER bound for the two kernels that gather records by
        # index; for those: hbm_bytes_calibrated = F + W + (the half of their STREAMED bytes the counter missed).
        R_avg = float(bj["config"].get("R_avg") or 0.0)
        if k.startswith("render_forward_kernel"):
            # streams: the point list, at most 4 R bytes from memory (the quarter waves of a tile re-read it from L2)
            entry["hbm_bytes_calibrated"] = f + w + 2.0 * R_avg
            entry["calibration"] = "F + W + 2 R_avg: record gathers count in full, the point-list stream (<= 4 R bytes) at one half"
        elif k.startswith("hb_fill_kernel"):
            # streams: 8 bytes per coarse pair (key, id); Rc ~ 0.22 R; tables are a few MB
            entry["hbm_bytes_calibrated"] = f + w + 4.0 * 0.22 * R_avg
            entry["calibration"] = "F + W + 4 Rc (Rc ~ 0.22 R_avg): one 64-byte record line per coarse pair counts in full, the pairs' stream at one half"
        else:
            entry["hbm_bytes_calibrated"] = 2.0 * f + w
        # instruction counters of the same kernel (other PMC passes), mean per dispatch: bench.py's
        # roofline_valu reads SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU (quad-cycles) of the render kernel
        for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES",
                  "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"):
            if c in agg[k]:
                entry[c] = sum(agg[k][c]) / len(agg[k][c])
        entry["duration_ns_under_pmc"] = sum(agg[k]["duration_ns"]) / len(agg[k]["duration_ns"])
        traffic[k] = entry
        # bare name = the variant the bench loop runs (the one with the most sampled launches)
        if k.startswith("render_forward_kernel") and (
                "render_forward_kernel" not in traffic or
                entry["launches_sampled"] > traffic["render_forward_kernel"]["launches_sampled"]):
            traffic["render_forward_kernel"] = dict(entry, variant=k)
json.dump(traffic, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)

# ---- config 5 (train): backward kernels ----
tstats = os.path.join(OUT, "prof_train_stats", "stats_kernel_stats.csv")
if os.path.exists(tstats):
    shutil.copy(tstats, os.path.join(dst, "%s_train_kernel_stats.csv" % tag))
    trows = list(csv.DictReader(open(tstats)))
    nb = [int(r["Calls"]) for r in trows if "render_backward" in r["Name"]]
    tl = ["# rocprofv3 --kernel-trace --stats of `python tools/bench_train.py --steps 10 --warmup 3`",
          "# (config 5: scene-149-like P = 1 M, 1920x1280, forward + backward; %d backward calls)" % (nb[0] if nb else 0),
          "", "%10s %8s  %s" % ("avg us", "calls", "kernel")]
    for r in sorted(trows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
        tl.append("%10.1f %8s  %s" % (float(r["AverageNs"]) / 1000, r["Calls"], r["Name"].split("(")[0][-80:]))
    tagg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(OUT, "prof_train_pmc*", "pmc_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("grpg::", "")
            tagg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tl += ["", "# PMC, mean per dispatch (FETCH_SIZE / WRITE_SIZE in KiB, raw)"]
    for k in sorted(tagg):
        if "backward" in k:
            tl.append("== %s" % k)
            for c, v in sorted(tagg[k].items()):
                tl.append("   %-24s %16.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
    open(os.path.join(dst, "%s_train_summary.txt" % tag), "w").write("\n".join(tl) + "\n")
    try:
        tb = open(os.path.join(OUT, "prof_train_bench.json")).read().strip().splitlines()[-1]
        open(os.path.join(dst, "%s_train_under_rocprof.json" % tag), "w").write(tb + "\n")
    except Exception:
        pass
print("\n".join(lines[:30]))
