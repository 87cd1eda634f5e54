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


def _trailing_cases():
    """a valid frame, then a frame header and a first block header the walk stops at: the trailing frame holds NO block"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import zgdata
    packs = read_pack("decodecorpus.pack")
    good_ck = packs["z000033.zst"]                                      # the corpus frames carry a content checksum ...
    good_nock = zgdata.zstd_compress(zgdata.text_like(300000, seed=5), checksum=False)  # ... this one has none
    cases = []
    for good in (good_ck, good_nock):
        for other in (good_ck, good_nock):
            hdr_len = 4 + 1 + (0 if (other[4] >> 5) & 1 else 1) + (0, 1, 2, 4)[other[4] & 3] + ((1 if (other[4] >> 5) & 1 else 0), 2, 4, 8)[other[4] >> 6]
            hdr = other[:hdr_len]
            cases.append(good + hdr)                                  # truncated behind the frame header
            cases.append(good + hdr + b"\x05")                        # ... inside the block header
            cases.append(good + hdr + bytes([0x06 | 1, 0, 0]))        # reserved block type (3), last block
            cases.append(good + hdr + (((200000 << 3) | 4 | 1).to_bytes(3, "little")))   # compressed block of 200000 bytes: too large
            cases.append(good + hdr + (((100 << 3) | 4).to_bytes(3, "little")) + b"xyz")   # body truncated
            cases.append(good + hdr[:hdr_len - 1])                     # the frame header itself truncated
    return cases


def test_trailing_frame_without_a_block():
    n = 0
    for m in _trailing_cases():
        ost, _ = oracle.FrameDecoder().decode_all(m, 1 << 25)
        assert ost != 0
        assert emu.decode_all_verdict(m) == ost, (n, ost, emu.decode_all_verdict(m))
        n += 1
    assert n == 24
