import sys, os
sys.path.insert(0,'zstd-rs_amd'); sys.path.insert(0,'tools')
os.environ['ZGPU_DEBUG_TIMERS']='1'
import zgdata, zgpu
d=zgdata.text_like(256<<20); z=zgdata.zstd_compress(d)
c=zgpu.Context(0, dev=True); b=c.prepare(z)
for _ in range(2): b.run(); b.sync()
t=b.debug_timers()
print("seq: decode ticks/WG", t[16]/max(t[18],1), "mover ticks/WG", t[17]/max(t[18],1), "WGs", t[18], b.timings())
