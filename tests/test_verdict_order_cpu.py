"""Inputs with more than one defect: the reference decodes block by block and sequence by sequence, so the FIRST defect in stream order
decides — whatever stage of the engine meets which defect first. The rules (found by the differential soaks tools/dev/soak.py on the GPU and
tools/dev/soak_cpu.py here): a defect inside a block comes before a header the host walk cannot read further back; a block's literal
verdicts come before what the host finds in its sequences section header; a sequence that reaches too far comes before a later sequence of
its block that zg_k_seqpost rejects (no literals left, offset 0). Checked on the CPU harness (host parser, verdict rules, zg_exact.h) against
the oracle: the fixtures the soaks found, and a small random soak."""
import glob
import os
import random

import emu
import oracle
from golden_io import read_pack

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fixtures_found_by_the_soaks():
    files = sorted(glob.glob(os.path.join(HERE, "golden", "verdict_order", "*.zst")))
    assert len(files) >= 4
    for f in files:
        m = open(f, "rb").read()
        ost, _ = oracle.FrameDecoder().decode_all(m, 1 << 25)
        assert ost != 0 and emu.decode_all_verdict(m) == ost, (os.path.basename(f), ost, emu.decode_all_verdict(m))


def test_small_random_soak():
    packs = read_pack("decodecorpus.pack")
    bases = [packs[n] for n in sorted(packs) if n.endswith(".zst")][::2]
    rng = random.Random(20260926)
    nerr = 0
    for bi, base in enumerate(bases):
        for it in range(3):
            m = bytearray(base)
            for _ in range(2 + rng.randrange(2)):
                if len(m) < 16:
                    break
                kind = rng.randrange(5)
                if kind == 0:
                    i = rng.randrange(4, len(m)); m[i] ^= 1 << rng.randrange(8)
                elif kind == 1:
                    i = rng.randrange(4, len(m)); m[i] = rng.randrange(256)
                elif kind == 2:
                    m = m[:rng.randrange(8, len(m))]
                elif kind == 3:
                    i = rng.randrange(4, len(m) - 4); m[i:i + 2] = bytes([rng.randrange(256), rng.randrange(256)])
                else:
                    i = rng.randrange(4, min(len(m), 40)); m[i] = rng.randrange(256)
            m = bytes(m)
            ost, _ = oracle.FrameDecoder().decode_all(m, 1 << 25)
            assert emu.decode_all_verdict(m) == ost, (bi, it, ost)
            nerr += 1 if ost else 0
    assert nerr > 80
