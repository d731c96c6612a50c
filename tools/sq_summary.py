"""Per-kernel summary of an SQ counter pass (tools/gpu.sh TAG pmcsq): where the wavefronts' cycles go and what an instruction costs.

    python tools/sq_summary.py gpurun_out/pmc_TAG/SQ [out.md]

SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts (MI355X_MICROARCH.md); WAIT_ANY (parked at s_waitcnt /
s_barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
launch = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xrhip::", "")
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launch[k].add(r["Dispatch_Id"])
out = ["| kernel | launches | wave quad-cycles / launch | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | VALU / SALU / LDS instructions per launch (all wavefronts) | quad-cycles of issue per instruction |",
       "|---|---|---|---|---|---|---|---|"]
for k, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = max(1, len(launch[k]))
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    ins = c.get("SQ_INSTS_VALU", 0) + c.get("SQ_INSTS_SALU", 0) + c.get("SQ_INSTS_LDS", 0)
    out.append("| `%s` | %d | %.0f | %.1f %% | %.1f %% | %.1f %% | %.0f / %.0f / %.0f | %.2f |" % (
        k, n, wc / n, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        c.get("SQ_INSTS_VALU", 0) / n, c.get("SQ_INSTS_SALU", 0) / n, c.get("SQ_INSTS_LDS", 0) / n,
        c.get("SQ_ACTIVE_INST_ANY", 0) / ins if ins else float("nan")))
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
