#!/usr/bin/env python3
"""rocprofv3 outputs of tools/dev/profile.sh (gpurun_out/prof/<workload>/...) -> the summaries kept under profiles/r06/.

  --reduce W..  (on the GPU box) condense the counter CSVs into gpurun_out/prof/pmc_reduced.json: per workload and kernel the summed
                counter and the number of launches; the SQ-counter CSVs are cut down to the engine's kernels (one row per launch
                and counter: the raw evidence kept under profiles/). The bulky traces stay behind.
  (default)     (here) write profiles/r06/: per workload <w>_kernel_stats.csv, <w>_bench_under_rocprof.json and <w>_pmc.json: HBM bytes
                per pass per kernel, raw and calibrated PER ACCESS PATTERN. On gfx950 FETCH_SIZE reports half of a wide coalesced
                read (MI355X_MICROARCH.md, HBM section); what it reports for the engine's other patterns is measured by the
                calibration kernels (zg_k_calib_*): the factor applied to a kernel is that of the pattern its reads are made of.
"""
import collections, csv, glob, hashlib, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles", "r06")
CSRC = os.path.join(ROOT, "zstd-rs_amd", "csrc")


def kname(s):
    return s.split("(")[0].replace("void ", "")


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = kname(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1].add(r["Dispatch_Id"])
    return {k: [v[0], len(v[1])] for k, v in acc.items()}


def find(d, suffix="counter_collection.csv"):
    for r, _, fs in os.walk(os.path.join(P, d)):
        for f in fs:
            if f.endswith(suffix):
                return os.path.join(r, f)
    return None


def reduce(workloads):
    red = {"calib_fetch": per_kernel(find("calib_fetch"), "FETCH_SIZE"), "calib_write": per_kernel(find("calib_write"), "WRITE_SIZE"), "workloads": {}}
    for w in workloads:
        f, wr = find(w + "/fetch"), find(w + "/write")
        if not f or not wr:
            print("no counters for", w)
            continue
        red["workloads"][w] = {"fetch": per_kernel(f, "FETCH_SIZE"), "write": per_kernel(wr, "WRITE_SIZE")}
        ks = find(w + "/stats", "kernel_stats.csv")
        if ks:
            shutil.copy(ks, os.path.join(P, w + "_kernel_stats.csv"))
        for k in (1, 2, 3):
            sq = find("%s/sq%d" % (w, k))
            if not sq:
                continue
            with open(os.path.join(P, "%s_sq%d.csv" % (w, k)), "w", newline="") as fo:
                wtr = csv.writer(fo)
                wtr.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value"])
                for r in csv.DictReader(open(sq)):
                    k2 = kname(r["Kernel_Name"])
                    if k2.startswith("zg_k_"):
                        wtr.writerow([r["Dispatch_Id"], k2, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", ""),
                                      r.get("VGPR_Count", ""), r.get("SGPR_Count", ""), r["Counter_Name"], r["Counter_Value"]])
    json.dump(red, open(os.path.join(P, "pmc_reduced.json"), "w"), indent=1)


# which calibration pattern a kernel's reads are (mostly) made of; its writes are 4 or 16 B per lane coalesced
READ_PATTERN = {"zg_k_sweep": "mixed_sweep", "zg_k_flatten": "mixed_flat", "zg_k_flatten4": "mixed_flat", "zg_k_seqpost": "copy16", "zg_k_seq": "copy16", "zg_k_huf": "copy16",
                "zg_k_ftab": "copy16", "zg_k_tables": "copy16", "zg_k_scan": "copy4", "zg_k_lit": "copy16"}


def main():
    if "--reduce" in sys.argv:
        return reduce([a for a in sys.argv[1:] if not a.startswith("-")])
    os.makedirs(OUT, exist_ok=True)
    red = json.load(open(os.path.join(P, "pmc_reduced.json")))
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
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.kernels_sha256()
    for w, rw in red["workloads"].items():
        ks = os.path.join(P, w + "_kernel_stats.csv")
        if os.path.exists(ks):
            shutil.copy(ks, os.path.join(OUT, w + "_kernel_stats.csv"))
        # SQ counters: per kernel and counter the sum over the run's launches and the number of launches (the raw per-launch CSVs of round 4
        # were 40,000 lines of build-specific numbers in git: ADVICE r4 — tools/dev/profile.sh regenerates them on demand under gpurun_out/)
        sq = {}
        for f in sorted(glob.glob(os.path.join(P, w + "_sq*.csv"))):
            for r in csv.DictReader(open(f)):
                if not kname(r["Kernel_Name"]).startswith("zg_k_"):
                    continue
                e = sq.setdefault(kname(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], [0, 0])
                e[0] += int(float(r["Counter_Value"])); e[1] += 1
        if sq:
            json.dump({"kernels_sha256": sha, "workload": w, "note": "rocprofv3 --pmc SQ_* passes of bench.py --workload %s --steps 1 --warmup 1 (three passes of counters): "
                       "[sum over the run's launches, launches]" % w, "counters": sq}, open(os.path.join(OUT, w + "_sq_summary.json"), "w"))
        lines = [l for l in open(os.path.join(P, w, "stats.log")) if l.startswith('{"metric"')]
        if not lines:
            print("no bench line for", w)
            continue
        bench = json.loads(lines[-1])
        json.dump(bench, open(os.path.join(OUT, w + "_bench_under_rocprof.json"), "w"), indent=1)
        D = bench["config"]["plaintext_bytes_job"]
        plan = bench.get("lz77_plan", {})
        Dp, Dd = plan.get("pointer_bytes", D), plan.get("direct_bytes", 0)
        res = {"kernels_sha256": sha, "workload": w, "plaintext_bytes": D, "lz77_plan": plan, "calibration": cal,
               "method": "FETCH_SIZE / WRITE_SIZE (KiB) from separate rocprofv3 --pmc passes, summed over all launches of a kernel and divided by the "
                         "passes of the run. Reads: a kernel whose reads are wide coalesced loads gets the factor measured on zg_k_calib_copy "
                         "(16 B/lane); narrow coalesced loads that of zg_k_calib_copy4; random 4/8-byte gathers that of zg_k_calib_gather (64 B per "
                         "miss assumed true). zg_k_sweep and zg_k_flatten mix a known amount of wide reads (sweep: the scratch words of the "
                         "pointer-mode units, 4 B per byte; flatten: sequence records, ~1.2 B per byte of a unit with sequences) with gathers: "
                         "the wide part is corrected analytically, the rest keeps the gather factor.",
               "kernels": {}}
        once = [n for k, (v, n) in rw["fetch"].items() if k.split("<")[0] in ("zg_k_scan", "zg_k_seq", "zg_k_ftab")]
        npass = min(once) if once else 4
        total = 0.0
        for k in sorted(set(rw["fetch"]) | set(rw["write"])):
            if not k.startswith("zg_k_") or "calib" in k:
                continue
            f = rw["fetch"].get(k, [0.0, 1]); wv = rw["write"].get(k, [0.0, 1])
            fraw, wraw = f[0] * 1024.0 / npass, wv[0] * 1024.0 / npass             # bytes per pass
            base = k.split("<")[0]
            pat = READ_PATTERN.get(base, "copy16")
            if pat == "copy16":
                fcor = fraw * f16
            elif pat == "copy4":
                fcor = fraw * f4
            else:
                wide_true = 4.0 * Dp if base == "zg_k_sweep" else 1.2 * (Dp + Dd)
                wide_raw = min(wide_true / f16, fraw)
                fcor = wide_raw * f16 + (fraw - wide_raw) * fg
            wcor = wraw * w16
            res["kernels"][k] = {"launches_per_pass": round(f[1] / npass, 2), "fetch_raw_bytes_per_pass": int(fraw), "write_raw_bytes_per_pass": int(wraw),
                                 "read_pattern": pat, "hbm_bytes_per_pass_calibrated": int(fcor + wcor)}
            total += fcor + wcor
        res["passes_in_run"] = npass
        res["pipeline_hbm_bytes_per_pass"] = int(total)
        res["algorithmic_bytes_per_pass"] = bench["roofline"]["algorithmic_bytes"]
        json.dump(res, open(os.path.join(OUT, w + "_pmc.json"), "w"), indent=1)
        print("== %s (%d passes)" % (w, npass))
        for k, v in res["kernels"].items():
            print("%-34s x%-7.2f raw fetch %8.3f GB  raw write %8.3f GB  calibrated %8.3f GB  (%s)" % (k, v["launches_per_pass"], v["fetch_raw_bytes_per_pass"] / 1e9,
                  v["write_raw_bytes_per_pass"] / 1e9, v["hbm_bytes_per_pass_calibrated"] / 1e9, v["read_pattern"]))
        print("pipeline %.3f GB per pass; algorithmic %.3f GB" % (total / 1e9, res["algorithmic_bytes_per_pass"] / 1e9))


if __name__ == "__main__":
    main()
