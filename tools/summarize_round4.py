"""gpurun_out/prof_<tag>/ (tools/profile_round4.sh) -> profiles/<tag>_<workload>_{kernel_stats.csv, pmc_fetch_size.csv, pmc_write_size.csv,
pmc_sq.csv, bench_line.json} and profiles/counters.json.

counters.json is keyed by the SHAPE of the run -- the bench line's config.shape: workload, pairs, length, mode, bandwidth, scoring -- and,
below that, by the profiler's kernel name; bench.py prints roofline.traffic / roofline.issue only for a shape that has an entry (no scaling
of one shape's counts to another).  Units as the MI355X guide prescribes: FETCH_SIZE / WRITE_SIZE are in KB, each collected in a pass of
its own; bench.py doubles FETCH_SIZE (gfx950 tallies 64 of every 128 bytes of a wide read), WRITE_SIZE was calibrated on the kernels' own
store pattern (profiles/r03_write_size_calibration.txt).

usage: python tools/summarize_round4.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    """{kernel name: {counter: (sum over dispatches, dispatches)}}"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            a = agg[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1].add(row["Dispatch_Id"])
    return {k: {c: (v[0], len(v[1])) for c, v in cs.items()} for k, cs in agg.items()}


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    cpath = os.path.join(dst, "counters.json")
    counters = json.load(open(cpath)) if os.path.exists(cpath) else {}
    for line in sorted(glob.glob(os.path.join(src, "*_bench_line.json"))):
        name = os.path.basename(line)[:-len("_bench_line.json")]
        if not os.path.getsize(line):
            continue
        cfg = json.loads(open(line).read().strip().splitlines()[-1])
        shutil.copy(line, os.path.join(dst, "%s_%s_bench_line.json" % (tag, name)))
        for f in glob.glob(os.path.join(src, "**", name + "_kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
        shape = cfg["config"].get("shape")
        if not shape:
            continue
        ent = {}
        for ctr, field, scale in (("FETCH_SIZE", "fetch_raw_bytes_per_launch", 1024.0), ("WRITE_SIZE", "write_bytes_per_launch", 1024.0),
                                  ("SQ", None, 1.0)):
            fs = glob.glob(os.path.join(src, "**", "%s_pmc_%s_counter_collection.csv" % (name, ctr)), recursive=True)
            if not fs:
                continue
            pk = per_kernel(fs[0])
            with open(os.path.join(dst, "%s_%s_pmc_%s.csv" % (tag, name, ctr.lower())), "w") as f:      # a per-kernel digest, not the per-dispatch dump
                f.write("Kernel_Name,Counter,Dispatches,Total,Per_dispatch\n")
                for k, cs in sorted(pk.items()):
                    for c, (tot, nd) in sorted(cs.items()):
                        f.write('"%s",%s,%d,%.1f,%.1f\n' % (k, c, nd, tot, tot / max(nd, 1)))
            for k, cs in pk.items():
                e = ent.setdefault(k, {})
                if field:
                    tot, nd = cs.get(ctr, (0.0, 0))
                    if nd:
                        e[field] = tot * scale / nd
                else:
                    for c, key in (("SQ_INSTS_VALU", "valu_per_launch"), ("SQ_INSTS_SALU", "salu_per_launch"), ("SQ_WAVE_CYCLES", "wave_cycles_per_launch"),
                                   ("SQ_WAIT_ANY", "wait_any_per_launch")):
                        tot, nd = cs.get(c, (0.0, 0))
                        if nd:
                            e[key] = tot / nd
        keep = {}
        for k, e in ent.items():
            if e.get("valu_per_launch", 0) + e.get("salu_per_launch", 0) < 1e7 and e.get("write_bytes_per_launch", 0) + e.get("fetch_raw_bytes_per_launch", 0) < 1e8:
                continue          # utility kernels
            e["source"] = "profiles/%s_%s_pmc_*.csv (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* in separate passes of the same bench command, tools/profile_round4.sh)" % (tag, name)
            keep[k] = e
        if keep:
            counters[shape] = keep
    json.dump(counters, open(cpath, "w"), indent=1, sort_keys=True)
    print("shapes in profiles/counters.json:", len(counters))
    for s, e in counters.items():
        print(" ", s, "->", ", ".join(sorted(e)))


if __name__ == "__main__":
    main()
