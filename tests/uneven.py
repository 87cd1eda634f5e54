"""Test helper: re-split the four Huffman streams of a literals section. libzstd always splits (regen + 3) / 4 symbols per stream
(the format's rule); ruzstd decodes the streams one after the other into one buffer and only compares the total
(literals_section_decoder.rs:94-155), so any split is valid for it. This takes a frame libzstd made (one compressed block whose
literals are Huffman-coded in four streams), decodes it with the oracle to get the code table and the literals, and writes the
same literals back with a different split."""
import oracle


def resplit(z, deltas, regen_bias=0, spare_bits=(0, 0, 0, 0)):
    """z: a frame of ONE compressed block with 4-stream Huffman literals (no dictionary). deltas: how many symbols to move:
    stream k gets (regen + 3) // 4 + deltas[k] symbols (k < 3), the fourth stream the rest. regen_bias: added to the section's
    Regenerated_Size field (a wrong total); spare_bits[k]: zero bits put below stream k's last code (a stream that does not end on
    its last bit). Returns (new frame, symbols per stream)."""
    o = oracle.FrameDecoder()
    st, c, _, _ = o.init(z)
    assert st == 0
    st, _, fin = o.decode_blocks(z[c:], oracle.STRAT_UPTO_BLOCKS, 1)
    assert st == 0
    lits = o.last_literals()
    entries, max_bits = o.huf_table()                 # [(symbol, num_bits)] by state
    code = {}
    for idx, (sym, nb) in enumerate(entries):
        code.setdefault(sym, (idx >> (max_bits - nb), nb))
    # ---- parse: block header, literals section header (compressed, 4 streams)
    bh = int.from_bytes(z[c:c + 3], "little")
    assert (bh >> 1) & 3 == 2
    bsize = bh >> 3
    body = z[c + 3:c + 3 + bsize]
    tail = z[c + 3 + bsize:]
    b0 = body[0]
    assert b0 & 3 == 2, "literals are not Huffman-compressed with a new table"
    sf = (b0 >> 2) & 3
    assert sf >= 1, "single stream"
    hlen = {1: 3, 2: 4, 3: 5}[sf]
    bits = {1: 10, 2: 14, 3: 18}[sf]
    h = int.from_bytes(body[:hlen], "little") >> 4
    regen, comp = h & ((1 << bits) - 1), h >> bits
    assert regen == len(lits)
    pay = body[hlen:hlen + comp]
    rest = body[hlen + comp:]
    hb = pay[0]
    desc = 1 + hb if hb < 128 else 1 + (hb - 127 + 1) // 2
    # ---- the new streams
    seg = (regen + 3) // 4
    counts = [seg + deltas[0], seg + deltas[1], seg + deltas[2]]
    counts.append(regen - sum(counts))
    assert all(n > 0 for n in counts)
    streams, at = [], 0
    for n in counts:
        acc = 1
        for s in lits[at:at + n]:
            cd, nb = code[s]
            acc = (acc << nb) | cd
        acc <<= spare_bits[len(streams)]
        streams.append(acc.to_bytes((acc.bit_length() + 7) // 8, "little"))
        at += n
    jumps = b"".join(len(s).to_bytes(2, "little") for s in streams[:3])
    newpay = pay[:desc] + jumps + b"".join(streams)
    assert len(newpay) < (1 << bits)
    nh = (2 | (sf << 2)) | (((regen + regen_bias) | (len(newpay) << bits)) << 4)
    newbody = nh.to_bytes(hlen, "little") + newpay + rest
    nbh = (len(newbody) << 3) | (bh & 7)
    return z[:c] + nbh.to_bytes(3, "little") + newbody + tail, counts
