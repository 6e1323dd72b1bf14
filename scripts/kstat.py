"""Average duration of the kernels whose name contains a pattern, from a rocprofv3 --stats output directory.
usage: kstat.py <dir> <pattern> [...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(p in r["Name"] for p in sys.argv[2:]):
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:9.2f} us')
