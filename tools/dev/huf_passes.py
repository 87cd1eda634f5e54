#!/usr/bin/env python3
"""How many passes zg_k_huf's windows take (CPU emulator, tests/emu built with -DZG_HUF_STATS=1): windows, passes per window, wave
steps per window (a pass lasts as long as its busiest lane), lane steps per symbol. usage: huf_passes.py [iso|text] [bytes]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import zgdata
kind = sys.argv[1] if len(sys.argv) > 1 else "iso"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
emu_dir = os.path.join(ROOT, "tests", "emu")
so = "/tmp/libzg_emu_stats.so"
open("/tmp/zg_huf_stats.c", "w").write("unsigned long long zg_huf_stats[8];\n")
subprocess.check_call("g++ -O2 -std=c++17 -fPIC -shared -DZG_HUF_STATS=1 %s -Wno-unknown-pragmas -fno-strict-aliasing -o %s zg_emu.cpp zg_emu_flat.cpp zg_emu_exact.cpp zg_emu_huf.cpp "
                      "../../zstd-rs_amd/csrc/zg_host_parse.cpp /tmp/zg_huf_stats.c" % (os.environ.get("EXTRA", ""), so), shell=True, cwd=emu_dir)
L = C.CDLL(so)
L.zgemu_decode3.restype = C.c_void_p
L.zgemu_decode3.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32]
L.zgemu_huf.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
L.zgemu_lit_bytes.restype = C.c_uint64; L.zgemu_lit_bytes.argtypes = [C.c_void_p]
L.zgemu_num_blocks.argtypes = [C.c_void_p]
plain = zgdata.iso_like(n, seed=0x150) if kind == "iso" else zgdata.text_like(n, seed=0xE9)
z = zgdata.zstd_compress(plain)
h = L.zgemu_decode3(z, len(z), 1 << 31, 1, 0, 0)
nb = L.zgemu_num_blocks(h)
lb = L.zgemu_lit_bytes(h)
lit = np.zeros(lb + 64, dtype=np.uint8); st = np.zeros(nb + 1, dtype=np.uint32); cnt = np.zeros(4 * nb + 4, dtype=np.uint32)
L.zgemu_huf(h, 0, lit.ctypes.data, None, st.ctypes.data, cnt.ctypes.data)
s = (C.c_ulonglong * 8).in_dll(L, "zg_huf_stats")
w, passes, wave_steps, lane_steps, redo = s[0], s[1], s[2], s[3], s[4]
print("%s %d bytes: literals %d, windows %d, passes per window %.2f, wave steps per window %.1f (per symbol and wave %.3f), lane steps per symbol %.2f, lanes redone per window %.1f"
      % (kind, n, lb, w, passes / max(w, 1), wave_steps / max(w, 1), wave_steps / max(lb, 1), lane_steps / max(lb, 1), redo / max(w, 1)))
