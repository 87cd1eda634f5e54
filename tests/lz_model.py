"""numpy model of what zg_k_flatten must leave in its scratch (test infrastructure).

For every output byte of a unit (a run of consecutive blocks of one frame): 0 for a literal byte (and every byte of a
raw / RLE block), else the effective offset e with byte[pos] = byte[pos - e], where pos - e is a literal byte or lies
before the unit. Computed from the oracle's sequences (ll, ml, actual offset) by following parents until the chain hits
a literal or leaves the unit."""
import ctypes as C

import numpy as np

import oracle


def frame_parents(z):
    """decode one frame block by block with the oracle. Returns (parent[], block_starts[]): parent[x] = position byte x
    copies from (x - offset), or -1 for a literal byte / a byte of a raw or RLE block"""
    d = oracle.FrameDecoder()
    st, c, _, _ = d.init(z)
    assert st == 0
    L = d.L
    pos, out, chunks, starts = c, 0, [], []
    while True:
        hdr = z[pos] | (z[pos + 1] << 8) | (z[pos + 2] << 16)       # Block_Header: last(1) type(2) size(21)
        st, used, fin = d.decode_blocks(z[pos:], oracle.STRAT_UPTO_BLOCKS, 1)
        assert st == 0
        pos += used
        starts.append(out)
        if d.last_block_type() == 2:
            n, nl = C.c_size_t(), C.c_size_t()
            p = L.zor_last_sequences(d.h, C.byref(n))
            L.zor_last_literals(d.h, C.byref(nl))
            if n.value:
                a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value, 4)).astype(np.int64)
                ll, ml, of = a[:, 0], a[:, 1], a[:, 3]
                size = int(nl.value + ml.sum())
                par = np.full(size, -1, dtype=np.int64)
                mstart = np.cumsum(ll + ml) - ml
                tot = int(ml.sum())
                idx = np.repeat(mstart, ml) + (np.arange(tot) - np.repeat(np.cumsum(ml) - ml, ml))
                par[idx] = out + idx - np.repeat(of, ml)
            else:
                size = int(nl.value)
                par = np.full(size, -1, dtype=np.int64)
        else:
            size = hdr >> 3
            par = np.full(size, -1, dtype=np.int64)
        chunks.append(par)
        out += size
        if fin or d.is_finished():
            break
    return (np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64)), starts


def expected_scratch(z, unit_first_blocks):
    """unit_first_blocks: frame-relative index of the first block of every unit. Returns (e[], unit_starts[] + [N])"""
    parent, starts = frame_parents(z)
    N = len(parent)
    bounds = [starts[b] for b in unit_first_blocks] + [N]
    x = np.arange(N, dtype=np.int64)
    ustart = np.zeros(N, dtype=np.int64)
    for i in range(len(bounds) - 1):
        ustart[bounds[i]:bounds[i + 1]] = bounds[i]
    lit = parent < 0
    ptr = np.where(lit | (parent < ustart), x, parent)
    while True:
        nxt = ptr[ptr]
        if np.array_equal(nxt, ptr):
            break
        ptr = nxt
    src = np.where(lit[ptr], ptr, parent[ptr])
    return np.where(lit, 0, x - src).astype(np.uint32), bounds
