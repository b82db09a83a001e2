"""Timeline of the LAST fit in a rocprofv3 kernel trace CSV: per-kernel totals, per-stream busy time,
idle gaps on the panel stream, and the update kernels' duration vs. remaining size."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# split into fits at each gram kernel
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void agp::gram_") and "diag" not in r["Kernel_Name"] or "gram_fast" in r["Kernel_Name"]]
first = starts[-1]
fit = rows[first:]
t0 = fit[0]["s"]
print("fit span ms", (max(r["e"] for r in fit) - t0) / 1e6, "kernels", len(fit))
tot = collections.defaultdict(lambda: [0, 0])
for r in fit:
    k = r["Kernel_Name"].split("(")[0][:70]
    tot[k][0] += r["e"] - r["s"]; tot[k][1] += 1
for k, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{d/1e6:9.3f} ms {c:5d}  {k}")
qkey = "Queue_Id" if "Queue_Id" in fit[0] else None
streams = collections.defaultdict(list)
for r in fit:
    streams[(r.get("Queue_Id"), r.get("Stream_Id"))].append(r)
for q, rs in streams.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print("queue/stream", q, "kernels", len(rs), "busy ms", busy / 1e6, "from", (rs[0]["s"] - t0) / 1e6, "to", (rs[-1]["e"] - t0) / 1e6)
if len(sys.argv) > 2:
    for r in fit:
        print(f'{(r["s"]-t0)/1e3:10.1f} {(r["e"]-r["s"])/1e3:8.1f} q{r.get("Queue_Id")} s{r.get("Stream_Id")} g{r.get("Grid_Size_X", r.get("Grid_Size"))} {r["Kernel_Name"][:60]}')
