"""Reader/writer for the packed golden fixtures under tests/golden/.

A pack is: b"ZGPK" u32 count, then per entry: u16 name_len, name (utf-8), u32 data_len, data.
The packs hold the reference's own golden inputs (its .zst fixtures, dictionary and fuzz
artefacts) so that GPU-box tests never read /root/reference; the expected plaintexts are
kept as (size, sha256) in the JSON manifests written by make_golden.py.
"""
import json
import os
import struct

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def write_pack(path, entries):
    with open(path, "wb") as f:
        f.write(b"ZGPK" + struct.pack("<I", len(entries)))
        for name, data in entries:
            nb = name.encode()
            f.write(struct.pack("<H", len(nb)) + nb + struct.pack("<I", len(data)) + data)


def read_pack(name):
    path = os.path.join(GOLDEN_DIR, name)
    with open(path, "rb") as f:
        blob = f.read()
    assert blob[:4] == b"ZGPK"
    (count,) = struct.unpack_from("<I", blob, 4)
    pos = 8
    out = {}
    for _ in range(count):
        (nl,) = struct.unpack_from("<H", blob, pos)
        pos += 2
        nm = blob[pos:pos + nl].decode()
        pos += nl
        (dl,) = struct.unpack_from("<I", blob, pos)
        pos += 4
        out[nm] = blob[pos:pos + dl]
        pos += dl
    return out


def read_manifest(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)
