"""Generate tests/golden/cli/: FASTA inputs and the REAL reference CLI's stdout for them (oracle/_ref/bsalign_ref_cli,
compiled from /root/reference/main.c by oracle/Makefile).  Run in the build container:  python tests/golden/make_golden_cli.py
tests/test_cli_gpu.py replays the same command lines through bsalign_amd/bsalign-hip and compares byte for byte."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import support as S  # noqa: E402

OUT = os.path.join(HERE, "cli")
REF = os.path.join(S.ROOT, "oracle", "_ref", "bsalign_ref_cli")


def fasta(path, recs, width=70, fastq=False):
    with open(path, "w") as f:
        for name, seq in recs:
            if fastq:
                f.write("@%s\n%s\n+\n%s\n" % (name, seq, "I" * len(seq)))
            else:
                f.write(">%s some description\n" % name)
                for i in range(0, len(seq), width):
                    f.write(seq[i:i + width] + "\n")


def s(a):
    return "".join("ACGT"[int(c)] for c in a)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20240611)
    # C1 of BASELINE.json: two 1 kbp sequences, global, bandwidth 64
    T = rng.integers(0, 4, size=1000).astype(np.uint8)
    Q = S.mutate(rng, T, 0.10)
    fasta(os.path.join(OUT, "c1.fa"), [("read_q", s(Q)), ("read_t", s(T))])
    # several pairs, lower-case and an N, different lengths
    recs = []
    for k in range(4):
        t = rng.integers(0, 4, size=int(rng.integers(150, 700))).astype(np.uint8)
        q = S.mutate(rng, t, float(rng.choice([0.05, 0.15])))
        qs, ts = s(q), s(t)
        if k == 1:
            qs = qs.lower()
        if k == 2:
            ts = ts[:40] + "N" + ts[41:]
        recs += [("q%d" % k, qs), ("t%d" % k, ts)]
    fasta(os.path.join(OUT, "multi.fa"), recs)
    fasta(os.path.join(OUT, "multi.fq"), recs[:4], fastq=True)
    cases = [
        ("c1_global_w64", ["align", "-m", "global", "-W", "64", "c1.fa"]),
        ("c1_default", ["align", "c1.fa"]),
        ("c1_global_w64_lines", ["align", "-m", "global", "-W", "64", "-L", "1", "c1.fa"]),
        ("c1_extend_2piece", ["align", "-m", "extend", "-W", "128", "-M", "2", "-X", "6", "-O", "3", "-E", "2", "-Q", "8", "-P", "1", "c1.fa"]),
        ("multi_overlap", ["align", "-W", "96", "multi.fa"]),
        ("multi_global_paper", ["align", "-m", "global", "-W", "128", "-M", "2", "-X", "2", "-O", "4", "-E", "2", "multi.fa"]),
        ("multi_fastq", ["align", "-m", "global", "-W", "64", "multi.fq"]),
        ("c1_edit", ["edit", "c1.fa"]),
        ("c1_edit_w128", ["edit", "-W", "128", "c1.fa"]),
        ("multi_edit_overlap", ["edit", "-m", "overlap", "multi.fa"]),
        ("multi_edit_extend", ["edit", "-m", "extend", "multi.fa"]),
        ("c1_edit_kmer", ["edit", "-m", "kmer", "c1.fa"]),
        ("multi_edit_kmer_k9", ["edit", "-m", "kmer", "-k", "9", "multi.fa"]),
    ]
    manifest = []
    for name, args in cases:
        r = subprocess.run([REF] + args, cwd=OUT, capture_output=True, timeout=120)
        assert r.returncode == 0, (name, r.stderr[-300:])
        with open(os.path.join(OUT, name + ".out"), "wb") as f:
            f.write(r.stdout)
        manifest.append({"name": name, "args": args})
        print(name, len(r.stdout), "bytes")
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
