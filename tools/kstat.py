import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-70s calls %6s  avg %9.1f us  per-step %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / steps))
print("sum per step (all kernels): %.1f us" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e3 / steps))
