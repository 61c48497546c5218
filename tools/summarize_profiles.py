"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the small files kept under profiles/ and refresh
profiles/hbm_traffic.json (HBM bytes per launch of each workload's dominant kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of the
bytes of wide coalesced reads, so FETCH_SIZE is doubled).

usage: python tools/summarize_profiles.py <tag>"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = {"align8": "k_align8_fwd", "edit": "k_edit_fwd", "poa": "k_sweep", "editfull": "k_edit_fwd_wide",
            "align8wq": "k_align8_fwd_sys", "align8wq4096": "k_align8_fwd_sys", "poarec": "k_poa_wf", "poarec4096": "k_poa_wf"}


def pmc_sum(path, kernel_prefix):
    """-> (sum of counter values over dispatches of the kernel, number of dispatches)"""
    tot, disp = 0.0, set()
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel_prefix in row["Kernel_Name"]:
                tot += float(row["Counter_Value"])
                disp.add(row["Dispatch_Id"])
    return tot, len(disp)


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    traffic = {}
    tpath = os.path.join(dst, "hbm_traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
    for wl, kern in DOMINANT.items():
        for f in glob.glob(os.path.join(src, "**", wl + "_kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, wl)))
        line = os.path.join(src, wl + "_bench_line.json")
        cfg = None
        if os.path.exists(line) and os.path.getsize(line):
            shutil.copy(line, os.path.join(dst, "%s_%s_bench_line.json" % (tag, wl)))
            cfg = json.loads(open(line).read().strip().splitlines()[-1])
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            fs = glob.glob(os.path.join(src, "**", "%s_pmc_%s_counter_collection.csv" % (wl, ctr)), recursive=True)
            if not fs:
                continue
            tot, nd = pmc_sum(fs[0], kern)
            vals[ctr] = (tot * 1024.0, nd)
            # keep a per-kernel digest, not the raw per-dispatch dump
            out = os.path.join(dst, "%s_%s_pmc_%s.csv" % (tag, wl, ctr.lower()))
            agg = {}
            with open(fs[0]) as f:
                for row in csv.DictReader(f):
                    k = row["Kernel_Name"]
                    a = agg.setdefault(k, [0, 0.0])
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
            with open(out, "w") as f:
                f.write("Kernel_Name,Dispatches,%s_KB_total,%s_KB_per_dispatch\n" % (ctr, ctr))
                for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    f.write('"%s",%d,%.1f,%.1f\n' % (k, n, v, v / n))
        if cfg and "FETCH_SIZE" in vals and "WRITE_SIZE" in vals and vals["FETCH_SIZE"][1]:
            c = cfg["config"]
            n = c.get("pairs_per_gpu", c.get("windows_per_gpu"))
            L = c.get("length", c.get("positions"))
            key = "%s_n%d_L%d_bw%d" % (c["workload"].split(":")[0] if ":" in c.get("workload", "") else wl, n, L, c["bandwidth"])
            nd = vals["FETCH_SIZE"][1]
            fetch, write = vals["FETCH_SIZE"][0] / nd, vals["WRITE_SIZE"][0] / vals["WRITE_SIZE"][1]
            traffic[key] = {"kernel": kern, "dispatches_measured": nd, "bytes_per_launch": 2.0 * fetch + write,
                            "fetch_size_raw_bytes_per_launch": fetch, "write_size_bytes_per_launch": write,
                            "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (profiles/%s_%s_pmc_*.csv); counters in KB; "
                                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads)" % (tag, wl)}
    for name in ("kmer_kernel_stats.csv", "kmer_timing.log"):
        for f in glob.glob(os.path.join(src, "**", name), recursive=True):
            shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, name)))
    json.dump(traffic, open(tpath, "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
