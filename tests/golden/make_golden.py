#!/usr/bin/env python3
"""Pack the reference's golden fixtures into tests/golden/ (run in the build container).

Inputs (read-only): /root/reference/ruzstd/{decodecorpus_files,dict_tests,test_fixtures,fuzz/artifacts}
Outputs: *.pack (the compressed inputs, verbatim) and *.json manifests holding, for every
input with a known plaintext, its size and sha256 (the plaintexts themselves are not copied).
These are the vectors the reference's own tests pin.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from golden_io import write_pack  # noqa: E402

REF = os.environ.get("ZGPU_REFERENCE", "/root/reference/ruzstd")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def rd(p):
    with open(p, "rb") as f:
        return f.read()


def pack_pairs(src_dir, pack_name, manifest_name, extra=()):
    entries, manifest = [], {}
    for fn in sorted(os.listdir(src_dir)):
        if not fn.endswith(".zst"):
            continue
        z = rd(os.path.join(src_dir, fn))
        plain = rd(os.path.join(src_dir, fn[:-4]))
        entries.append((fn, z))
        manifest[fn] = {"zst_size": len(z), "size": len(plain), "sha256": sha(plain)}
    for nm, data in extra:
        entries.append((nm, data))
    write_pack(os.path.join(HERE, pack_name), entries)
    with open(os.path.join(HERE, manifest_name), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    return len(manifest)


def main():
    n1 = pack_pairs(os.path.join(REF, "decodecorpus_files"), "decodecorpus.pack", "decodecorpus.json")
    dict_raw = rd(os.path.join(REF, "dict_tests", "dictionary"))
    n2 = pack_pairs(os.path.join(REF, "dict_tests", "files"), "dict_tests.pack", "dict_tests.json",
                    extra=[("dictionary", dict_raw)])
    # window fixtures: plaintexts are formulas (ruzstd/src/tests/mod.rs:582-595)
    fx = os.path.join(REF, "test_fixtures")
    fox = b"The quick brown fox jumps over the lazy dog.\n" * 4096
    sphinx = b"Sphinx of black quartz, judge my vow.\n" * 4096
    entries, manifest = [], {}
    for fn, plain in (("window_8mib.zst", sphinx), ("window_128mib.zst", fox), ("window_256mib.zst", fox)):
        z = rd(os.path.join(fx, fn))
        entries.append((fn, z))
        manifest[fn] = {"zst_size": len(z), "size": len(plain), "sha256": sha(plain)}
    entries.append(("abc.txt.zst", rd(os.path.join(fx, "abc.txt.zst"))))
    write_pack(os.path.join(HERE, "test_fixtures.pack"), entries)
    with open(os.path.join(HERE, "test_fixtures.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    # fuzz artefacts: must-not-crash inputs (ruzstd/src/tests/fuzz_regressions.rs:2-27)
    entries = []
    art = os.path.join(REF, "fuzz", "artifacts")
    for sub in sorted(os.listdir(art)):
        for fn in sorted(os.listdir(os.path.join(art, sub))):
            entries.append((sub + "/" + fn, rd(os.path.join(art, sub, fn))))
    write_pack(os.path.join(HERE, "fuzz_artifacts.pack"), entries)
    # synthetic inputs compressed by the image's libzstd (SURVEY.md Appendix C generators, tools/zgdata.py):
    # real-encoder streams (all-FSE modes, 4-stream Huffman, Treeless/Repeat lineage) small enough to commit
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools"))
    import zgdata
    syn, sman = [], {}
    cases = [("text_1m_l3.zst", zgdata.text_like(1 << 20), 3), ("text_1m_l1.zst", zgdata.text_like(1 << 20, seed=0xEA), 1),
             ("text_768k_l19.zst", zgdata.text_like(768 << 10, seed=0xEB), 19), ("iso_512k_l3.zst", zgdata.iso_like(512 << 10), 3),
             ("mixed_640k_l3.zst", zgdata.text_like(256 << 10, seed=7) + zgdata.iso_like(128 << 10, seed=9) + bytes(64 << 10) + zgdata.text_like(192 << 10, seed=7), 3)]
    for nm, plain, lvl in cases:
        z = zgdata.zstd_compress(plain, level=lvl)
        syn.append((nm, z))
        sman[nm] = {"zst_size": len(z), "size": len(plain), "sha256": sha(plain), "level": lvl, "zstd": zgdata.zstd_version()}
    write_pack(os.path.join(HERE, "synthetic.pack"), syn)
    with open(os.path.join(HERE, "synthetic.json"), "w") as f:
        json.dump(sman, f, indent=0, sort_keys=True)
    print("packed", n1, "corpus pairs,", n2, "dict pairs, 4 fixtures,", len(entries), "fuzz artefacts")


if __name__ == "__main__":
    main()
