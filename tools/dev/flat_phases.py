import sys, os
sys.path.insert(0,'zstd-rs_amd'); sys.path.insert(0,'tools')
os.environ['ZGPU_DEBUG_TIMERS']='1'
import zgdata, zgpu
d=zgdata.text_like(256<<20); z=zgdata.zstd_compress(d)
c=zgpu.Context(0); b=c.prepare(z)
for _ in range(2): b.run(); b.sync()
t=b.debug_timers(); tot=sum(t[:5])+sum(t[8:12])
print("flat phases (cycles summed over WGs): sync0,S1,S2,S3,S4+store:", [round(x/tot,3) for x in t[:5]], "total Mcycles", tot/1e6, b.timings())
print("rounds/tile", t[5]/max(t[7],1), "tiles", t[7], "byte-rounds", t[8], "match bytes", t[9], "unresolved after flat", t[10], "of", len(d))
print("S1 loop (wave0)", t[11]/tot, "S1 rest", t[1]/tot, "S2", t[2]/tot, "S3 prep", t[8]/tot, "S3 gather wait", t[9]/tot, "S3 finish", t[10]/tot, "S3 barrier", t[3]/tot, "S4", t[4]/tot, "sync0", t[0]/tot)
