"""Per-launch times of ONE network forward out of a rocprofv3 kernel trace (bench.py --streams 1): every convolution /
finish / stem / coordinate-plan launch of the last complete scene, in launch order, with its grid and resources.
usage: python profiles/layer_trace.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::|cvsc::|void ", "", r["Kernel_Name"]).split("(")[0]
starts = [i for i, r in enumerate(rows) if "sort_minmax" in r["Kernel_Name"]]        # first launch of a scene's coordinate plan
assert len(starts) >= 3, "expected several scenes in the trace (found %d sort_minmax launches)" % len(starts)
a, b = starts[-2], starts[-1]
scene = rows[a:b]
t0 = int(scene[0]["Start_Timestamp"])
print("# one scene, one in flight: %d launches, %.1f us from the first launch to the end of the last" % (
    len(scene), (int(scene[-1]["End_Timestamp"]) - t0) / 1e3))
print("%4s %9s %8s %8s %9s %5s %6s %5s  %s" % ("#", "start_us", "dur_us", "gap_us", "grid", "wg", "lds", "vgpr", "kernel"))
prev = t0
tot = {}
for i, r in enumerate(scene):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = name(r)
    tot[n] = tot.get(n, [0, 0.0])
    tot[n][0] += 1
    tot[n][1] += (e - s) / 1e3
    print("%4d %9.1f %8.1f %8.1f %9s %5s %6s %5s  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size", ""),
                                                       r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), n))
    prev = e
print("# totals per kernel (launches, us):")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("#   %-40s %4d %9.1f" % (n, c, t))
