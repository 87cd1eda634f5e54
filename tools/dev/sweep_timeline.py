"""Timeline of the sweep launches of the last pass in a rocprofv3 --kernel-trace CSV: which launches overlap.
usage: sweep_timeline.py <kernel_trace.csv>   (tails: small grids on the main stream; heads: large grids on the side stream)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "zg_k_sweep" in r["Kernel_Name"] or "zg_k_flatten" in r["Kernel_Name"] or "zg_k_fin" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last pass = after the last flat
last = max(i for i, r in enumerate(rows) if "zg_k_flatten" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["End_Timestamp"])
import collections
sig = lambda r: (r.get("Grid_Size_X", r.get("Grid_Size", "0")), r.get("Grid_Size_Y", "1"))
sw = [r for r in rows if "sweep" in r["Kernel_Name"]]
tail_sig = collections.Counter(sig(r) for r in sw).most_common(1)[0][0]    # the chain: one launch per step, all of one shape
tails = [r for r in sw if sig(r) == tail_sig]
heads = [r for r in sw if sig(r) != tail_sig]
print("tails %d  heads %d  queues %s" % (len(tails), len(heads), sorted(set(r.get("Queue_Id", "?") for r in rows))))
for r in heads:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    inside = [x for x in tails if int(x["Start_Timestamp"]) >= int(r["Start_Timestamp"]) and int(x["End_Timestamp"]) <= int(r["End_Timestamp"])]
    di = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in inside]
    print("head  start %8.1f us  end %8.1f us  dur %7.1f us  tails inside: %d (avg %.1f us each)" % (s, e, e - s, len(inside), sum(di) / max(len(di), 1)))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tails]
g = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(tails, tails[1:])]
print("tails: avg dur %.2f us, avg gap %.2f us, first start %.1f us, last end %.1f us" % (sum(d) / len(d), sum(g) / max(len(g), 1), (int(tails[0]["Start_Timestamp"]) - t0) / 1e3, (int(tails[-1]["End_Timestamp"]) - t0) / 1e3))
fin = [r for r in rows if "zg_k_fin" in r["Kernel_Name"]]
if fin: print("fin at %.1f us" % ((int(fin[-1]["Start_Timestamp"]) - t0) / 1e3))
