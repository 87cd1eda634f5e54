#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/dev/profile.sh (gpurun_out/prof) into the summaries kept under profiles/r01/:
kernel-trace statistics (copied), HBM bytes per launch per kernel from the FETCH_SIZE / WRITE_SIZE passes (KiB units),
with the fetch correction calibrated on zg_k_calib_copy (1 GiB read + 1 GiB written), and the bench line of the traced run."""
import csv, json, os, shutil, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles", "r01")


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in acc.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    shutil.copy(os.path.join(P, "stats", "trace_kernel_stats.csv"), os.path.join(OUT, "bench_1e9_kernel_stats.csv"))
    line = [l for l in open(os.path.join(P, "stats.log")) if l.startswith('{"metric"')][-1]
    json.dump(json.loads(line), open(os.path.join(OUT, "bench_1e9_under_rocprof.json"), "w"), indent=1)
    cf = per_kernel(os.path.join(P, "calib_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")["zg_k_calib_copy"]
    cw = per_kernel(os.path.join(P, "calib_write", "pmc_counter_collection.csv"), "WRITE_SIZE")["zg_k_calib_copy"]
    copy = 1 << 30
    fcorr = copy / (cf[0] / cf[1] * 1024.0)
    wcorr = copy / (cw[0] / cw[1] * 1024.0)
    F = per_kernel(os.path.join(P, "fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
    W = per_kernel(os.path.join(P, "write", "pmc_counter_collection.csv"), "WRITE_SIZE")
    res = {"calibration": {"copy_bytes": copy, "FETCH_SIZE_KiB": cf[0] / cf[1], "WRITE_SIZE_KiB": cw[0] / cw[1],
                           "fetch_correction": fcorr, "write_correction": wcorr,
                           "note": "zg_k_calib_copy reads and writes exactly 1 GiB with 16 B per lane; FETCH_SIZE reports half of it on gfx950 "
                                   "(MI355X_MICROARCH.md HBM section), WRITE_SIZE is exact. Both raw and corrected totals are given per kernel."},
           "workload": "bench.py --size 1000000000 (text_like 1e9 B | zstd -3, one frame)", "kernels": {}}
    for k in sorted(set(F) | set(W)):
        if not k.startswith("zg_k_"):
            continue
        f = F.get(k, (0.0, 1)); w = W.get(k, (0.0, 1))
        fk, wk = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
        res["kernels"][k] = {"FETCH_SIZE_KiB_per_launch": round(fk, 1), "WRITE_SIZE_KiB_per_launch": round(wk, 1),
                             "hbm_bytes_per_launch_corrected": int(fk * 1024 * fcorr + wk * 1024 * wcorr),
                             "hbm_bytes_per_launch_raw": int((fk + wk) * 1024)}
    json.dump(res, open(os.path.join(OUT, "bench_1e9_pmc.json"), "w"), indent=1)
    for k, v in res["kernels"].items():
        print("%-16s fetch %10.1f MiB  write %10.1f MiB  corrected %.3f GB" % (k, v["FETCH_SIZE_KiB_per_launch"] / 1024, v["WRITE_SIZE_KiB_per_launch"] / 1024, v["hbm_bytes_per_launch_corrected"] / 1e9))


if __name__ == "__main__":
    main()
