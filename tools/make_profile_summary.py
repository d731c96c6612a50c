"""Turns the rocprofv3 outputs of one profiling call into the committed summaries under profiles/.

On the GPU box (one gpurun call; PMC passes are separate runs without any other trace domain):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/prof_TAG -o full -- python $R/bench.py --steps 100 --warmup 40 --cpu-frames 0 --no-profile
    for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_TAG/$C -o pmc -- python $R/bench.py --steps 60 --warmup 40 --cpu-frames 0 --no-profile; done
    python $R/bench.py > $R/gpurun_out/bench_TAG.json
(tools/gpu.sh TAG tests bench trace peaks pmc is that command sequence.)  Here:
    python tools/make_profile_summary.py TAG v13 "one-line description of this version"
XR_ROUND (default r01) is the prefix of the files written under profiles/ (r02 in round 2, ...); the PMC header text is
taken from the newest existing r*_pmc_traffic.md.
"""
import collections
import csv
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, ver, desc = sys.argv[1], sys.argv[2], sys.argv[3]
rnd = os.environ.get("XR_ROUND", "r01")
go = os.path.join(ROOT, "gpurun_out")

import glob


def first(*patterns):
    """the first existing path among the glob patterns (tools/gpu.sh numbers its outputs by step: prof_TAG_3, bench_TAG_2.json)"""
    for pat in patterns:
        hits = sorted(glob.glob(os.path.join(go, pat)))
        if hits:
            return hits[0]
    raise SystemExit("nothing matches " + " / ".join(patterns))


c = sqlite3.connect(first("prof_%s/full_results.db" % tag, "prof_%s_*/full_results.db" % tag))
rows = c.execute("select name,start,end from kernels order by start").fetchall()
tot = collections.defaultdict(lambda: [0, 0])
durs = collections.defaultdict(list)   # per kernel: every launch's duration (ns) -- an average hides one-offs (km_jacobi: one 3.9 ms run + 26 empty ones)
for n, s, e in rows:
    k = n.split("(")[0]
    if k.startswith("void "):
        k = k[5:]
    tot[k][0] += e - s
    tot[k][1] += 1
    durs[k].append(e - s)
total = sum(t for t, _ in tot.values())
b = [ln for ln in open(os.environ.get("XR_BENCH_JSON") or first("bench_%s.json" % tag, "bench_%s_[0-9]*.json" % tag)) if ln.startswith("{")][-1].strip()
bj = json.loads(b)
frames = max(1, tot["xrhip::k_clahe_lut"][1])   # one CLAHE pass per camera frame: counts the frames the traced command processed
lines = ["# round %d, full pipeline %s (%s)" % (int(rnd[1:]), ver, desc), "",
         "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 40 --cpu-frames 0 --variant-frames 0 --sustained-frames 0 --no-profile` (the reference-shaped call: inline, host image) on one MI355X",
         "(gfx950, ROCm 7.2); kernel-trace statistics from the results database.", "",
         ("Total kernel time %.2f ms over %d frames (%.3f ms / frame); default `python bench.py` on the same box: %.1f frames/s, "
          "%.4f ms/frame, %.4f ms/BA-iteration, CPU reference %.1f frames/s on 1 core (`" + rnd + "_full_%s_bench.json`).") %
         (total / 1e6, frames, total / 1e6 / frames, bj["value"], bj["ms_per_step"], bj["ms_per_ba_iteration"],
          bj.get("cpu_baseline", {}).get("value", float("nan")), ver), "",
         "| kernel | calls | total ms | avg us | min / median / max us | % |", "|---|---|---|---|---|---|"]
for k, (t, n) in sorted(tot.items(), key=lambda x: -x[1][0]):
    d = sorted(durs[k])
    lines.append("| `%s` | %d | %.2f | %.2f | %.1f / %.1f / %.1f | %.1f |" % (k, n, t / 1e6, t / 1e3 / n, d[0] / 1e3, d[len(d) // 2] / 1e3, d[-1] / 1e3,
                                                                   100 * t / total))
open(os.path.join(ROOT, "profiles", "%s_full_%s_kernel_stats.md" % (rnd, ver)), "w").write("\n".join(lines) + "\n")
# the same table for bench.py (`device_busy_frac`, `kernel_us_per_frame` of the line -- shown beside timings of the same kernel revision only)
json.dump({"_kernel_rev": bj.get("kernel_rev"), "_workload": os.environ.get("XR_WORKLOAD", "s1"), "_frames": frames,
           "_source": "%s_full_%s_kernel_stats.md" % (rnd, ver), "kernel_ms_per_frame": round(total / 1e6 / frames, 4),
           "kernels": {k.replace("xrhip::", ""): {"calls": n, "avg_us": round(t / 1e3 / n, 2), "us_per_frame": round(t / 1e3 / frames, 2)}
                       for k, (t, n) in sorted(tot.items(), key=lambda x: -x[1][0])}},
          open(os.path.join(ROOT, "profiles", "%s_kernel_stats.json" % rnd), "w"), indent=1)
open(os.path.join(ROOT, "profiles", "%s_full_%s_bench.json" % (rnd, ver)), "w").write(b + "\n")

pmc_dir = os.path.join(go, "pmc_%s" % tag)
if os.path.isdir(pmc_dir):
    res = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(lambda: [0.0, 0])
        with open(os.path.join(pmc_dir, C, "pmc_counter_collection.csv")) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != C:
                    continue
                n = r["Kernel_Name"].split("(")[0].replace("xrhip::", "")
                if n.startswith("void "):
                    n = n[5:]
                n = n.split("<")[0]
                agg[n][0] += float(r["Counter_Value"])
                agg[n][1] += 1
        res[C] = agg
    names = sorted(res["FETCH_SIZE"], key=lambda n: -res["FETCH_SIZE"][n][0])
    out = {}
    import glob
    md = os.path.join(ROOT, "profiles", "%s_pmc_traffic.md" % rnd)
    head = open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.md")))[-1]).read().split("| kernel |")[0]
    head = "# round %d (%s, kernel revision %s): HBM-side traffic per launch from rocprofv3 PMC counters\n" % (
        int(rnd[1:]), ver, bj.get("kernel_rev")) + head.split("\n", 1)[1]
    head += "| kernel | launches | FETCH_SIZE KB/launch | WRITE_SIZE KB/launch |\n|---|---|---|---|\n"
    for n in names:
        f = res["FETCH_SIZE"][n]
        w = res["WRITE_SIZE"].get(n, [0, 1])
        fk, wk = f[0] / f[1], w[0] / max(1, w[1])
        out[n] = {"launches": f[1], "fetch_kb": round(fk, 2), "write_kb": round(wk, 2)}
        head += "| `%s` | %d | %.1f | %.1f |\n" % (n, f[1], fk, wk)
    open(md, "w").write(head)
    out["_kernel_rev"] = bj.get("kernel_rev")   # bench.py shows these numbers only beside timings of the same kernels ...
    out["_workload"] = os.environ.get("XR_WORKLOAD", "s1")   # ... on the same workload (a per-launch figure depends on the problems)
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % rnd), "w"), indent=1)
print("\n".join(lines[5:22]))
print(bj["roofline"])
print(bj["roofline_lk"])

# MFMA utilisation pass (SQ counters, own run): per kernel the f64 MFMA work the counters saw against the time the kernel
# took.  SQ_INSTS_VALU_MFMA_MOPS_F64 counts units of 512 flop (a v_mfma_f64_16x16x4_f64 is 4 units);
# SQ_VALU_MFMA_BUSY_CYCLES are cycles a SIMD's matrix pipe was busy, summed over the chip.
mf = os.path.join(pmc_dir, "MFMA", "pmc_counter_collection.csv")
if not os.path.exists(mf):
    mf = os.path.join(pmc_dir, "MFMA_ALT", "pmc_counter_collection.csv")
if os.path.exists(mf):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    with open(mf) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"].split("(")[0].replace("xrhip::", "")
            if n.startswith("void "):
                n = n[5:]
            agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], n)
            if key not in seen:
                seen.add(key)
                agg[n]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                agg[n]["_launches"] += 1
    peaks, peak_src = {}, None
    try:
        peaks = json.load(open(os.path.join(go, "peaks_%s.json" % tag)))
        json.dump(peaks, open(os.path.join(ROOT, "profiles", "%s_peaks.json" % rnd), "w"))
        peak_src = "measured in the same call (`tools/peaks.hip`, `%s_peaks.json`)" % rnd
    except (OSError, ValueError):
        old_peaks = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_peaks.json")))
        if old_peaks:       # the newest committed micro-benchmark of another box of the pool: say so
            peaks = json.load(open(old_peaks[-1]))
            peak_src = "measured on ANOTHER box of the pool (`%s`)" % os.path.basename(old_peaks[-1])
    if "mfma_f64_16x16x4_tflops" in peaks:
        pk, pk_label = peaks["mfma_f64_16x16x4_tflops"], "of measured peak"
    else:
        pk, pk_label, peak_src = 78.6, "of vendor peak", "NOT measured: the vendor figure"
    clock_ghz, simds = peaks.get("clock_mhz", 2400) / 1e3, peaks.get("cus", 256) * 4
    L = ["# round %d: f64 MFMA utilisation per kernel (%s)" % (int(rnd[1:]), ver), "",
         "`rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace` (own pass, no other",
         "trace domain) of `python bench.py --steps 60 --warmup 40 --cpu-frames 0 --no-profile`.  flop = MOPS_F64 x 512; TFLOP/s = flop / kernel",
         "duration of the same pass; `%s` against the f64 MFMA rate %.1f TFLOP/s -- %s;" % (pk_label, pk, peak_src),
         "the vendor figure is 78.6; `pipe busy` = SQ_VALU_MFMA_BUSY_CYCLES / (duration x %.1f GHz x %d SIMDs)." % (clock_ghz, simds), "",
         "| kernel | launches | avg us | MFMA flop / launch | TFLOP/s | %s | pipe busy |" % pk_label, "|---|---|---|---|---|---|---|"]
    out = {}
    for n, a in sorted(agg.items(), key=lambda x: -x[1].get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0)):
        mops = a.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
        if mops <= 0:
            continue
        fl, ns, nl = mops * 512.0, a["_ns"], a["_launches"]
        tf = fl / ns / 1e3
        busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (ns * clock_ghz * simds)
        out[n] = {"launches": int(nl), "avg_us": round(ns / nl / 1e3, 2), "mfma_flop_per_launch": round(fl / nl, 1),
                  "tflops": round(tf, 5), "frac_of_measured_peak": round(tf / pk, 7), "pipe_busy": round(busy, 7)}
        L.append("| `%s` | %d | %.2f | %.3g | %.4f | %.2e | %.2e |" % (n, nl, ns / nl / 1e3, fl / nl, tf, tf / pk, busy))
    open(os.path.join(ROOT, "profiles", "%s_mfma_util_%s.md" % (rnd, ver)), "w").write("\n".join(L) + "\n")
    json.dump(out, open(os.path.join(ROOT, "profiles", "%s_mfma_util.json" % rnd), "w"), indent=1)
