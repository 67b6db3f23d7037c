"""rocprofv3 --stats kernel table -> share / calls / average per kernel and the total per (step, layer): python tools/prof_table.py <kernel_stats.csv> <steps> <layers>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps, layers = int(sys.argv[2]), int(sys.argv[3])
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%5.1f%%  calls %5d  avg %8.1f us  per layer-step %7.1f us  %s" % (float(r["TotalDurationNs"]) / tot * 100, int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                                        float(r["TotalDurationNs"]) / steps / layers / 1e3, r["Name"][:100]))
print("kernel time per layer and step: %.1f us (gaps between launches not included)" % (tot / steps / layers / 1e3))
