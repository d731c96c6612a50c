"""Where does the frame time go between kernels?  Reads the rocprofv3 kernel-trace database of a profiling call
(gpurun_out/prof_TAG/full_results.db, written by tools/gpu.sh TAG trace) and reports, over the whole trace:

  * busy time (union of all kernel intervals -- kernels on the auxiliary streams overlap the main chain),
  * idle time between kernels, split by the length of the gap (dispatch gaps of a few microseconds inside a solve versus
    the host phases between solves),
  * for every kernel name: how long the device stayed idle after it before the next kernel started (median / mean), i.e.
    what a launch merged into its predecessor would save.

    python tools/kernel_gaps.py TAG [frames]          (frames: number of frames in the trace, default 140)
"""
import collections
import os
import sqlite3
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name,start,end from kernels order by start").fetchall()
    out = []
    for n, s, e in rows:
        k = n.split("(")[0]
        if k.startswith("void "):
            k = k[5:]
        out.append((k.replace("xrhip::", ""), int(s), int(e)))
    return out


def analyse(rows, frames):
    if not rows:
        return {}
    # union of busy intervals; remember which kernel ended last before every idle gap
    gaps = []          # (length ns, name of the kernel whose end opened the gap, name of the kernel that closed it)
    busy = 0
    cur_s, cur_e, cur_last = rows[0][1], rows[0][2], rows[0][0]
    for name, s, e in rows[1:]:
        if s <= cur_e:
            if e > cur_e:
                cur_e, cur_last = e, name
        else:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_last, name))
            cur_s, cur_e, cur_last = s, e, name
    busy += cur_e - cur_s
    span = max(e for _, _, e in rows) - rows[0][1]
    kernel_sum = sum(e - s for _, s, e in rows)
    buckets = collections.OrderedDict((("< 3 us", 0), ("3-10 us", 0), ("10-50 us", 0), ("50-500 us", 0), (">= 0.5 ms", 0)))
    counts = dict.fromkeys(buckets, 0)
    for g, _, _ in gaps:
        key = "< 3 us" if g < 3e3 else "3-10 us" if g < 1e4 else "10-50 us" if g < 5e4 else "50-500 us" if g < 5e5 else ">= 0.5 ms"
        buckets[key] += g
        counts[key] += 1
    after = collections.defaultdict(list)
    for g, last, _ in gaps:
        if g < 5e4:          # dispatch gaps; longer ones are host phases
            after[last].append(g)
    return {"span_ms": span / 1e6, "busy_ms": busy / 1e6, "kernel_sum_ms": kernel_sum / 1e6, "idle_ms": (span - busy) / 1e6,
            "frames": frames, "buckets": buckets, "counts": counts, "after": after}


def report(r):
    f = r["frames"]
    print("trace span %.2f ms, device busy (union) %.2f ms, sum of kernel times %.2f ms, idle %.2f ms" %
          (r["span_ms"], r["busy_ms"], r["kernel_sum_ms"], r["idle_ms"]))
    print("per frame (%d frames): busy %.3f ms, idle %.3f ms" % (f, r["busy_ms"] / f, r["idle_ms"] / f))
    print("\nidle time by gap length:")
    for k, v in r["buckets"].items():
        print("  %-10s %6d gaps  %8.3f ms  (%.3f ms / frame)" % (k, r["counts"][k], v / 1e6, v / 1e6 / f))
    print("\ndevice idle after a kernel until the next one starts (gaps < 50 us only):")
    print("  %-28s %7s %9s %9s %10s" % ("kernel", "gaps", "median us", "mean us", "ms / frame"))
    for k, v in sorted(r["after"].items(), key=lambda kv: -sum(kv[1])):
        print("  %-28s %7d %9.2f %9.2f %10.4f" % (k, len(v), statistics.median(v) / 1e3, statistics.mean(v) / 1e3, sum(v) / 1e6 / f))


if __name__ == "__main__":
    tag = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 140
    db = tag if tag.endswith(".db") else os.path.join(ROOT, "gpurun_out", "prof_%s" % tag, "full_results.db")
    report(analyse(load(db), frames))
