#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_pmc.sh (gpurun_out/pmc/) into
profiles/<round>/pmc/pmc_summary_<tag>.csv, copies the kernel-trace stats next to it and
refreshes profiles/hbm_traffic.json (HBM bytes per launch of the sweep / traceback kernels,
FETCH_SIZE and WRITE_SIZE corrected by the factors the calibration kernels give).

    python tools/pmc_summary.py <tag> [round]        e.g.  python tools/pmc_summary.py v5_mode1 r01
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "pmc")


def rows(pass_dir):
    for f in glob.glob(os.path.join(SRC, pass_dir, "*", "*counter_collection.csv")):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r["Kernel_Name"].split("(")[0], r["Counter_Name"], float(r["Counter_Value"])


def main():
    tag = sys.argv[1]
    rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
    out_dir = os.path.join(ROOT, "profiles", rnd, "pmc")
    os.makedirs(out_dir, exist_ok=True)
    summary = []
    means = {}
    for p in ("FETCH_SIZE", "WRITE_SIZE", "calib_FETCH_SIZE", "calib_WRITE_SIZE", "lds", "sq"):
        acc = defaultdict(list)
        for k, c, v in rows(p):
            acc[(k, c)].append(v)
        for (k, c), vs in sorted(acc.items()):
            if p.startswith(("FETCH", "WRITE")) and k.startswith("sw_"):
                vs = vs[2:] if len(vs) > 4 else vs          # drop the warm-up launches
            summary.append((p, k, c, len(vs), sum(vs) / len(vs), min(vs), max(vs)))
            means[(p, k, c)] = sum(vs) / len(vs)
    with open(os.path.join(out_dir, f"pmc_summary_{tag}.csv"), "w") as fh:
        fh.write("pass,kernel,counter,dispatches,mean,min,max\n")
        for r in summary:
            fh.write(",".join(str(x) for x in r) + "\n")
    for f in glob.glob(os.path.join(SRC, "trace", "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(ROOT, "profiles", rnd, f"{tag}_kernel_stats.csv"))

    # calibration: the kernels move 1.61 GB (read) / 1.61 GB (write) with one dword per lane
    cal_r = means.get(("calib_FETCH_SIZE", "read_dwords", "FETCH_SIZE"))
    cal_w = means.get(("calib_WRITE_SIZE", "write_dwords", "WRITE_SIZE"))
    bytes_moved = 1572864.0 * 1024.0                     # 1.61 GB, as tools/pmc_calib.hip prints
    f_fac = bytes_moved / (cal_r * 1024.0) if cal_r else 2.0
    w_fac = bytes_moved / (cal_w * 1024.0) if cal_w else 1.0

    def hbm(kernel):
        f = means.get(("FETCH_SIZE", kernel, "FETCH_SIZE"))
        w = means.get(("WRITE_SIZE", kernel, "WRITE_SIZE"))
        if f is None or w is None:
            return None
        return int(round((f * f_fac + w * w_fac) * 1024.0))      # counters are in KiB

    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 10 "
                     "--warmup 2 --no-cpu-baseline`, tools/profile_pmc.sh + tools/pmc_summary.py (" + tag + ")")
    doc["calibration"] = {"kernel": "tools/pmc_calib.hip (one dword per lane, 256 B per wave instruction, 1.61 GB)",
                          "FETCH_SIZE_factor": f_fac, "WRITE_SIZE_factor": w_fac}
    doc["fill_kernel"] = "sw_sweep_winmax_kernel (mode 1)"
    doc["fill_kernel_hbm_bytes_per_launch"] = hbm("sw_sweep_winmax_kernel")
    doc["traceback_kernel_hbm_bytes_per_launch"] = hbm("sw_traceback_winmax_kernel")
    # instructions of the sweep per pair (steady launches: the minimum over the dispatches leaves out the first, cold ones)
    mins = {(p_, k, c): mn for (p_, k, c, _n, _m, mn, _x) in summary}
    v = mins.get(("sq", "sw_sweep_winmax_kernel", "SQ_INSTS_VALU"))
    sa = mins.get(("sq", "sw_sweep_winmax_kernel", "SQ_INSTS_SALU"))
    if v and sa:
        doc["sweep_insts_per_pair_headline"] = {
            "valu": int(round(v / 1000.0)), "salu": int(round(sa / 1000.0)),
            "source": "SQ_INSTS_VALU / SQ_INSTS_SALU per launch / 1000 pairs (profiles/%s/pmc/pmc_summary_%s.csv), 150 x 2000 pairs only" % (rnd, tag)}
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc, indent=1))
    # gpurun MERGES what a call wrote into the local gpurun_out/: a second collection would be summarised together with the
    # first one's files.  The raw files are scratch once condensed.
    shutil.rmtree(SRC, ignore_errors=True)


if __name__ == "__main__":
    main()
