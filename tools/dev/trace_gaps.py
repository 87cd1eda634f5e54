"""Per-kernel durations and the idle gaps in front of each launch from a rocprofv3 --kernel-trace CSV: trace_gaps.py <kernel_trace.csv> [name substring]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.Counter()
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name] += e - s; cnt[name] += 1
    if prev_end is not None:
        gap[name] += max(0, s - prev_end)
    prev_end = max(prev_end or 0, e)
for n in sorted(dur, key=lambda n: -dur[n]):
    if pat in n:
        print("%-44s calls %6d  avg dur %9.2f us  avg gap before %7.2f us  total %9.3f ms" % (n[-44:], cnt[n], dur[n] / cnt[n] / 1e3, gap[n] / cnt[n] / 1e3, (dur[n] + gap[n]) / 1e6))
