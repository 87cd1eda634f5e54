import sys, os
sys.path.insert(0,'zstd-rs_amd'); sys.path.insert(0,'tools')
os.environ['ZGPU_DEBUG_TIMERS']='1'
import zgdata, zgpu
d=zgdata.text_like(256<<20); z=zgdata.zstd_compress(d)
c=zgpu.Context(0); b=c.prepare(z)
for _ in range(2): b.run(); b.sync()
t=b.debug_timers(); tot=sum(t[:5])
print("flat phases (cycles summed over WGs): sync0,S1,S2,S3,S4+store:", [round(x/tot,3) for x in t[:5]], "total Mcycles", tot/1e6, b.timings())
print("rounds/tile", t[5]/max(t[7],1), "tiles", t[7], "byte-rounds", t[8], "match bytes", t[9], "unresolved after flat", t[10], "of", len(d))
