#!/usr/bin/env python3
"""rocprofv3 outputs of tools/dev/profile.sh (gpurun_out/prof) -> the summaries kept under profiles/r02/.

  --reduce   (on the GPU box) condense the counter CSVs into gpurun_out/prof/pmc_reduced.json: per kernel the summed counter
             and the number of launches. The raw CSVs (one row per launch: hundreds of sweep launches) stay behind.
  (default)  (here) write profiles/r02/: kernel statistics, the bench line of the traced run, and bench_pmc.json: HBM bytes per
             pass per kernel, raw and calibrated PER ACCESS PATTERN. On gfx950 FETCH_SIZE reports half of a wide (16 B/lane)
             coalesced read (MI355X_MICROARCH.md, HBM section); what it reports for the engine's other patterns is measured by
             the calibration kernels (zg_k_calib_*): the factor applied to a kernel is that of the pattern its reads are made of.
"""
import collections, csv, hashlib, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles", "r02")


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1].add(r["Dispatch_Id"])
    return {k: [v[0], len(v[1])] for k, v in acc.items()}


def find(d):
    for r, _, fs in os.walk(os.path.join(P, d)):
        for f in fs:
            if f.endswith("counter_collection.csv"):
                return os.path.join(r, f)
    raise SystemExit("no counter csv under " + d)


def reduce():
    red = {"fetch": per_kernel(find("fetch"), "FETCH_SIZE"), "write": per_kernel(find("write"), "WRITE_SIZE"),
           "calib_fetch": per_kernel(find("calib_fetch"), "FETCH_SIZE"), "calib_write": per_kernel(find("calib_write"), "WRITE_SIZE")}
    json.dump(red, open(os.path.join(P, "pmc_reduced.json"), "w"), indent=1)
    for r, _, fs in os.walk(os.path.join(P, "stats")):
        for f in fs:
            if f.endswith("kernel_stats.csv"):
                shutil.copy(os.path.join(r, f), os.path.join(P, "kernel_stats.csv"))


# which calibration pattern a kernel's reads are (mostly) made of; its writes are 4 or 16 B per lane coalesced
READ_PATTERN = {"zg_k_sweep": "mixed_sweep", "zg_k_flat": "mixed_flat", "zg_k_seqpost": "copy16", "zg_k_seq": "copy16", "zg_k_huf": "copy16",
                "zg_k_ftab": "copy16", "zg_k_tables": "copy16", "zg_k_scan": "copy4", "zg_k_lit": "copy4"}


def main():
    if "--reduce" in sys.argv:
        return reduce()
    os.makedirs(OUT, exist_ok=True)
    red = json.load(open(os.path.join(P, "pmc_reduced.json")))
    shutil.copy(os.path.join(P, "kernel_stats.csv"), os.path.join(OUT, "bench_kernel_stats.csv"))
    line = [l for l in open(os.path.join(P, "stats.log")) if l.startswith('{"metric"')][-1]
    bench = json.loads(line)
    json.dump(bench, open(os.path.join(OUT, "bench_under_rocprof.json"), "w"), indent=1)
    GiB = 1 << 30
    cal = {}
    for name, true_r, true_w in (("zg_k_calib_copy", GiB, GiB), ("zg_k_calib_copy4", GiB, GiB), ("zg_k_calib_gather<unsigned int>", GiB // 64 * 64, 0),
                                 ("zg_k_calib_gather<unsigned long>", GiB // 64 * 64, 0)):
        f = red["calib_fetch"].get(name, [0, 1]); w = red["calib_write"].get(name, [0, 1])
        fk, wk = f[0] / max(f[1], 1) * 1024.0, w[0] / max(w[1], 1) * 1024.0
        cal[name] = {"true_read_bytes": true_r, "FETCH_SIZE_bytes": fk, "read_factor": (true_r / fk) if fk else None,
                     "true_write_bytes": true_w, "WRITE_SIZE_bytes": wk, "write_factor": (true_w / wk) if wk and true_w else None}
    # the gathers are random 4 / 8 byte reads that all miss: "true" = one 64-byte fabric request each (n = 1 GiB / 64 reads)
    f16 = cal["zg_k_calib_copy"]["read_factor"] or 2.0
    f4 = cal["zg_k_calib_copy4"]["read_factor"] or 1.0
    fg = cal["zg_k_calib_gather<unsigned long>"]["read_factor"] or 1.0
    w16 = cal["zg_k_calib_copy"]["write_factor"] or 1.0
    D = bench["config"]["plaintext_bytes_job"]
    res = {"kernels_sha256": hashlib.sha256(open(os.path.join(ROOT, "zstd-rs_amd", "csrc", "zg_kernels.hip"), "rb").read()).hexdigest(),
           "workload": bench["config"]["name"], "plaintext_bytes": D,
           "calibration": cal,
           "method": "FETCH_SIZE / WRITE_SIZE (KiB) from separate rocprofv3 --pmc passes. Reads: a kernel whose reads are wide coalesced loads gets "
                     "the factor measured on zg_k_calib_copy (16 B/lane); narrow coalesced loads that of zg_k_calib_copy4; random 4/8-byte gathers that "
                     "of zg_k_calib_gather (64 B per miss assumed true). zg_k_sweep and zg_k_flat mix a known amount of wide reads (scratch words / "
                     "sequence records) with gathers: the wide part is corrected analytically, the rest keeps the gather factor.",
           "kernels": {}}
    total = 0.0
    # launches per pass: every kernel but the sweep is launched once per pass
    npass = None
    for k, (v, n) in red["fetch"].items():
        if k.startswith("zg_k_flat"):
            npass = n
    for k in sorted(set(red["fetch"]) | set(red["write"])):
        if not k.startswith("zg_k_") or "calib" in k:
            continue
        f = red["fetch"].get(k, [0.0, 1]); w = red["write"].get(k, [0.0, 1])
        fraw, wraw = f[0] * 1024.0 / npass, w[0] * 1024.0 / npass             # bytes per pass
        base = k.split("<")[0]
        pat = READ_PATTERN.get(base, "copy16")
        if pat == "copy16":
            fcor = fraw * f16
        elif pat == "copy4":
            fcor = fraw * f4
        else:
            wide_true = 4.0 * D if base == "zg_k_sweep" else 1.2 * D      # sweep: the scratch, 4 B per output byte; flat: 12 B per sequence (~1.2 B/byte on text)
            wide_raw = wide_true / f16
            fcor = wide_true + max(fraw - wide_raw, 0.0) * fg
        wcor = wraw * w16
        res["kernels"][k] = {"launches_per_pass": round(f[1] / npass, 2), "fetch_raw_bytes_per_pass": int(fraw), "write_raw_bytes_per_pass": int(wraw),
                             "read_pattern": pat, "hbm_bytes_per_pass_calibrated": int(fcor + wcor)}
        total += fcor + wcor
    res["pipeline_hbm_bytes_per_pass"] = int(total)
    res["algorithmic_bytes_per_pass"] = bench["roofline"]["algorithmic_bytes"]
    json.dump(res, open(os.path.join(OUT, "bench_pmc.json"), "w"), indent=1)
    for k, v in res["kernels"].items():
        print("%-34s raw fetch %8.3f GB  raw write %8.3f GB  calibrated %8.3f GB  (%s)" % (k, v["fetch_raw_bytes_per_pass"] / 1e9, v["write_raw_bytes_per_pass"] / 1e9,
                                                                                  v["hbm_bytes_per_pass_calibrated"] / 1e9, v["read_pattern"]))
    print("pipeline %.3f GB per pass; algorithmic %.3f GB" % (total / 1e9, res["algorithmic_bytes_per_pass"] / 1e9))


if __name__ == "__main__":
    main()
