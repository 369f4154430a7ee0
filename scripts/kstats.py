"""prints calls / average us of the kernels whose names contain any of the given substrings, from a rocprofv3 kernel_stats.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2:]) or len(sys.argv) == 2:
        print("    %-64s calls %6s  avg %9.2f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
