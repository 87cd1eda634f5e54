"""The host's plan for the LZ77 stages (zg_host_parse.cpp, BatchBuilder::finish) on the CPU: how a submit is cut into units,
which units get a sweep step, which frames skip the sweep (zg_k_sparse). These are the invariants the kernels rely on."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import emu


def _inputs():
    try:
        import zgdata
        zgdata.libzstd()
    except Exception as e:  # pragma: no cover
        pytest.skip("libzstd not available to create inputs: %s" % e)
    text = [zgdata.text_like(n, seed=0x700 + i) for i, n in enumerate((5 << 20, 700000, 3 << 20))]
    iso = [zgdata.iso_like(n, seed=0x710 + i) for i, n in enumerate((6 << 20, 2 << 20))]
    frames = [text[0], iso[0], text[1], iso[1], text[2], b"", b"x" * 1000]
    return b"".join(zgdata.zstd_compress(p) for p in frames)


@pytest.mark.parametrize("slots,unit_blocks", [(256, 0), (8, 0), (256, 1), (256, 7)])
def test_units_steps_and_sparse_frames(slots, unit_blocks):
    p = emu.Plan(_inputs(), flat_slots=slots, unit_blocks=unit_blocks)
    covered = 0
    for f, (fb, nb, fu, nu, sf, sc, sparse) in enumerate(p.frames):
        # the frame's units partition its blocks, in order
        at = fb
        for u in range(fu, fu + nu):
            frame, first, n, noseq = p.units[u]
            assert frame == f and first == at and n >= 1
            assert bool(noseq & 1) == (not any(p.nseq[first:first + n]))
            # direct mode: the first unit of a frame that has sequences and is not sparse, and nothing else
            assert bool(noseq & 2) == (u == fu and not (noseq & 1) and not sparse and nu <= 32)
            if unit_blocks:
                assert n == unit_blocks or u == fu + nu - 1
            at += n
        assert at == fb + nb
        covered += nb
        # its blocks with sequences, as a range of the batch's list
        want = [b for b in range(fb, fb + nb) if p.nseq[b]]
        assert p.seq_blocks[sf:sf + sc] == want if sc else not want
        nsq = sum(p.nseq[fb:fb + nb])
        assert bool(sparse) == (nsq <= 2048 and nsq <= 4 * nb)
    assert covered == p.nblocks
    # sweep steps: every pointer-mode unit (it has sequences and is not its frame's first one) of a frame that is not sparse appears exactly once; a frame's units in
    # step order; no empty step; the lists back to back
    want_units = [u for u, (f, _, _, noseq) in enumerate(p.units) if not noseq and not p.frames[f][6]]
    assert sorted(p.step_units) == want_units
    off = 0
    last_step_of_frame = {}
    for i, (list_off, n, max_blocks) in enumerate(p.steps):
        assert list_off == off and n >= 1
        lst = p.step_units[off:off + n]
        assert max_blocks == max(p.units[u][2] for u in lst)
        assert len({p.units[u][0] for u in lst}) == n                      # at most one unit per frame and step
        for u in lst:
            f = p.units[u][0]
            k = u - p.frames[f][2]                                         # the unit's index in its frame
            assert last_step_of_frame.get(f, (-1, -1))[1] < k              # in frame order along the steps
            last_step_of_frame[f] = (i, k)
        off += n
    assert off == len(p.step_units)
