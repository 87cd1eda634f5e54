import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'zstd-rs_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
os.environ['ZGPU_DEBUG_TIMERS'] = '1'
import zgdata, zgpu
d = zgdata.text_like(int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20); z = zgdata.zstd_compress(d)
c = zgpu.Context(0); b = c.prepare(z)
for _ in range(2): b.run(); b.sync()
t = b.debug_timers(); tot = sum(t[:3]) or 1
print("sweep rank0 cycles: data(loads+stores issue), drain(syncthreads), barrier:", [round(x / tot, 3) for x in t[:3]], "total Mcycles", tot / 1e6, {k: round(v, 2) for k, v in b.timings().items()})
per = t[8:8 + 256]; xcc = t[520:520 + 256]
import collections
print("data cycles per WG: min %.2f avg %.2f max %.2f Mcyc" % (min(per) / 1e6, sum(per) / len(per) / 1e6, max(per) / 1e6))
srt = sorted(per); print("percentiles 50/90/99:", srt[128] / 1e6, srt[230] / 1e6, srt[253] / 1e6)
print("xcc histogram:", sorted(collections.Counter(xcc).items()))
