"""Summarise a rocprofv3 kernel_trace.csv: per-kernel mean duration, GPU busy union, dispatch rate."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows]
ev.sort()
# keep the second half (steady state)
t_lo = ev[len(ev) // 2][0]
ev = [e for e in ev if e[0] >= t_lo]
span = max(e[1] for e in ev) - ev[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"dispatches {len(ev)}  span {span/1e6:.2f} ms  busy-union {busy/1e6:.2f} ms ({100*busy/span:.1f}%)  "
      f"{span/len(ev)/1e3:.2f} us per dispatch  queues {len(set(e[3] for e in ev))}")
d = collections.defaultdict(list)
for s, e, k, _ in ev: d[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]].append(e - s)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"{k:50s} n={len(v):5d} mean={sum(v)/len(v)/1e3:8.1f} us  sum={sum(v)/1e6:8.2f} ms ({100*sum(v)/tot:.1f}%)")
