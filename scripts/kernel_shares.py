"""The kernels of a rocprofv3 --kernel-trace --stats run by share of kernel time.  usage: kernel_shares.py DIR [top=16]"""
import csv, glob, re, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
best = []
for x in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    r = list(csv.DictReader(open(x)))
    if len(r) > len(best):
        best = r
tot = sum(float(r["TotalDurationNs"]) for r in best)
for r in best[:top]:
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Name"])
    name = m.group(1) if m else r["Name"][:30]
    print("%-36s calls %6d avg %9.1f us  %5.1f%%" % (name, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.1f ms" % (tot / 1e6))
