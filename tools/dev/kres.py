"""Per-kernel register / LDS / scratch use of zg_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "zstd-rs_amd", "csrc", "zg_kernels.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", src, "-c", "-o", "/tmp/kres.o",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:], capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    else:
        rows[cur][k.split(" [")[0]] = v
print("%-34s %5s %5s %7s %6s %5s %7s" % ("kernel", "sgpr", "vgpr", "scratch", "spill", "occ", "lds"))
for k, r in rows.items():
    print("%-34s %5s %5s %7s %6s %5s %7s" % (k[-34:], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("ScratchSize"), r.get("VGPRs Spill"), r.get("Occupancy"), r.get("LDS Size")))
