#!/usr/bin/env python3
"""bench.py -- throughput of the banded striped DP hot path on MI355X.

One "step" = one pass of the hot path (stage -> forward DP -> traceback -> CIGAR compaction) over one batch
of synthetic read pairs that is already resident in HBM.  Default workload = BASELINE.json configs[1]:
100 k synthetic 10 kbp pairs, 8-bit, global, bandwidth 128, one GPU.  With --gpus N (launched by
torch.distributed.run, one rank per GPU) every rank aligns its own 100 k pairs (weak scaling, no data-path
collective: pairs are independent); RCCL is used only for the barrier / max-over-ranks of the timing.

Prints ONE JSON line on rank 0 (contract in the task statement): metric GCUPS, roofline of the dominant
kernel (forward DP) measured with HIP events inside the library, and a CPU baseline of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 1024               # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # maximum shader clock (MI355X_MICROARCH.md); the chip clocks to its power budget below it, so cycle figures derived from
                           # wall time are upper bounds of the cycles actually spent


def counters_for(config_workload, kernel):
    """the committed PMC passes of EXACTLY this configuration (profiles/counters.json, keyed by the bench line's config.workload string --
    pairs, length, mode, bandwidth, scoring -- and the profiler's kernel name; written by tools/summarize_round4.py from
    tools/profile_round4.sh).  A shape that was not profiled has no entry: no traffic, no issue fractions are printed for it."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    except Exception:
        return None
    ent = tab.get(config_workload)
    if not ent:
        return None
    fam = kernel.split()[0].split("<")[0]          # the library names a launch family ("k_align8_fwd_xq"), the profiler an instance ("k_align8_fwd_xq<16, 4, 1, true>")
    for kn, v in ent.items():
        if kn.split("<")[0] == fam:
            return dict(v, kernel_instance=kn)
    return None


def issue_fractions(ent, kernel_ms):
    """how densely the kernel issued: SIMD cycles (at the chip's 2.4 GHz maximum, MI355X_MICROARCH.md) per VALU / scalar instruction, from the
    instruction counts of the PMC pass of this very configuration and the launch's duration measured in this run over 1024 SIMDs.  The packed,
    three-operand and DPP instructions these kernels are made of occupy a SIMD for 4 cycles, plain two-operand 32-bit ones for 2
    (profiles/r02f_valu_rate_probe.txt), and a SIMD's turn at the CU's scalar unit comes every 4: `valu_frac` = 4 / cycles per VALU instruction, capped
    at 1 -- a stream with some 2-cycle instructions in it can issue faster than one per 4 cycles, which is then reported as full."""
    if not ent or kernel_ms <= 0 or "valu_per_launch" not in ent:
        return None
    cyc = SIMDS * kernel_ms * 1e-3 * CLOCK_HZ          # SIMD cycles of the chip during the launch
    cpv = cyc / max(ent["valu_per_launch"], 1.0)
    cps = cyc / max(ent["salu_per_launch"], 1.0)
    return {"simd_cycles_per_valu": round(cpv, 3), "simd_cycles_per_salu": round(cps, 3),
            "valu_frac": round(min(1.0, 4.0 / cpv), 4), "salu_frac": round(min(1.0, 4.0 / cps), 4),
            "valu_per_launch": ent["valu_per_launch"], "salu_per_launch": ent["salu_per_launch"], "clock_hz": CLOCK_HZ,
            "kernel_instance": ent.get("kernel_instance"), "counts_from": ent.get("source")}


def traffic_of(ent):
    """HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE doubled: gfx950 counts 64 of every 128 bytes of a wide read)"""
    if not ent or "fetch_raw_bytes_per_launch" not in ent or "write_bytes_per_launch" not in ent:
        return None
    return 2.0 * ent["fetch_raw_bytes_per_launch"] + ent["write_bytes_per_launch"]
SEED = 20240611


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="align8", choices=["align8", "edit", "poa"])
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU (default: 100000 for align8, 32768 for edit -- 210 GB of row planes, four waves per SIMD; poa: POA windows, 16384)")
    ap.add_argument("--length", type=int, default=0, help="target length (default 10000 / 100000; poa: graph positions per window, 10000)")
    ap.add_argument("--bw", type=int, default=0, help="bandwidth (default 128 / 256 / 128); -1 = the reference's bandwidth 0, the whole query")
    ap.add_argument("--mode", default="global", choices=["global", "overlap", "extend"], help="pairwise workloads only (the headline configurations are global)")
    ap.add_argument("--eps", type=float, default=0.10)
    ap.add_argument("--scoring", default="2,-6,-3,-2,0,0", help="M,X,O,E,Q,P (reference CLI defaults, main.c:264)")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs of the CPU baseline sample (0 = auto, -1 = skip)")
    ap.add_argument("--workspace-gb", type=float, default=0.0)
    ap.add_argument("--poa-source", default="auto", choices=["auto", "synthetic", "recorded", "fixture"], help="poa workload: graph programs recorded from the reference's end_bspoa on "
                    "synthetic reads (needs oracle/_ref; the default when it is there), the programs the reference recorded into tests/golden/poa_graph.npz tiled over the windows "
                    "(fixture: the default without oracle/_ref), or synthetic row-task programs for the first form of the sweep (poa_synth)")
    ap.add_argument("--poa-case", type=int, default=0, help="poa workload, fixture source: which case of tests/golden/poa_graph.npz (0: default parameters; 1, 2: global / extend; "
                    "3: one-piece gaps; 4: linear gaps, bandwidth 32; 5: bandwidth 64; 6: bandwidth 256)")
    ap.add_argument("--cpu-worker", default="", help=argparse.SUPPRESS)
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, count them over the process group, run the shard exchange on a small synthetic "
                    "batch and print {n_gpus, exchange}: the launcher's own test (gloo when there is no GPU)")
    ap.add_argument("--no-secondary", action="store_true", help="default run only: do not measure the edit (C3) and POA (C4-shaped) workloads beside the headline one")
    ap.add_argument("--no-exchange", action="store_true", help="skip the scatter / align / gather self-check of the shard exchange after the timed region")
    return ap.parse_args()


def ensure_ranks(args):
    """`--gpus N` is a promise about the number of ranks.  Started by torch.distributed.run (WORLD_SIZE set) the world
    size must equal N; started plainly with N > 1 this process replaces itself by that launcher, one rank per GPU,
    rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    ws = os.environ.get("WORLD_SIZE", "")
    if ws:
        if int(ws) != args.gpus:
            sys.exit("bench.py: --gpus %d but the launcher started %s ranks" % (args.gpus, ws))
        return
    if args.gpus <= 1 or args.cpu_worker:
        return
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def init_ranks(args):
    """-> (rank, world, local, dist or None); the process group is RCCL (backend "nccl") on a GPU box, gloo without one
    (launch check only).  The number of ranks the group itself counts is what the JSON line reports as n_gpus."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    have_gpu = torch.cuda.is_available()
    if have_gpu:
        if local >= torch.cuda.device_count():
            sys.exit("bench.py: rank %d has no GPU (%d visible)" % (local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
    if world <= 1:
        return rank, 1, local, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if have_gpu:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    one = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local) if have_gpu else "cpu")
    dist.all_reduce(one)
    seen = int(one.item())
    if seen != args.gpus:
        sys.exit("bench.py: --gpus %d but the process group counts %d ranks" % (args.gpus, seen))
    return rank, seen, local, dist


def exchange_selfcheck(dist, rank, world, dev, batch, bw, align=None, expect=None):
    """The one exchange each way of SURVEY.md 8(e) on a real batch, timed: rank 0 scatters contiguous shards (bsalign_amd/shard.py: lengths as
    broadcasts, shards packed by bsa_shard_pack, one grouped set of point-to-point messages), every rank produces its shard's records --
    align(shard) -> (results [n, 10] int32, cigar words, offsets [n + 1] int64) on its GPU, or a deterministic stand-in derived from the shard's own
    lengths when there is no GPU (launch check over gloo) --, rank 0 gathers them in pair order and compares with `expect` (its own results of the
    whole batch).  -> dict for the JSON line (rank 0), None elsewhere.  With one rank the group calls degenerate to local copies."""
    import torch
    from bsalign_amd import shard
    if dist is None:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        tdist.init_process_group("nccl" if dev.type == "cuda" else "gloo", rank=0, world_size=1, **({"device_id": dev} if dev.type == "cuda" else {}))
        own = True
        dist = tdist
    else:
        own = False
    sync = (lambda: torch.cuda.synchronize()) if dev.type == "cuda" else (lambda: None)
    sync(); dist.barrier(); sync()
    t0 = time.perf_counter()
    sh = shard.scatter_batch(batch if rank == 0 else None, bw, src=0, device=dev)
    sync(); dist.barrier(); sync()
    t1 = time.perf_counter()
    n = len(sh["qlen"])
    if align is not None:
        res, cig, off = align(sh)
    else:
        # no GPU: records that only depend on the pair -- (global index, qlen, tlen, ...) and qlen % 7 + 1 words per pair
        gi = np.arange(n, dtype=np.int64) + sh["first"]
        r = np.zeros((n, 10), np.int32); r[:, 0] = gi; r[:, 1] = sh["qlen"]; r[:, 2] = sh["tlen"]
        cnt = (sh["qlen"].astype(np.int64) % 7) + 1
        offh = np.zeros(n + 1, np.int64); offh[1:] = np.cumsum(cnt)
        words = (np.repeat(gi, cnt) * 16 + 1).astype(np.int32)
        res, cig, off = torch.from_numpy(r).to(dev), torch.from_numpy(words).to(dev), torch.from_numpy(offh).to(dev)
    sync(); dist.barrier(); sync()
    t2 = time.perf_counter()
    got = shard.gather_batch(res, cig, off, dst=0)
    sync(); dist.barrier(); sync()
    t3 = time.perf_counter()
    out = None
    if rank == 0:
        gr, gw, go = got[0].cpu().numpy(), got[1].cpu().numpy().view(np.uint32), got[2].cpu().numpy()
        if expect is not None:
            er, ew, eo = expect
            same = bool(np.array_equal(gr, er)) and bool(np.array_equal(go, eo)) and bool(np.array_equal(gw[: int(go[-1])], ew[: int(eo[-1])]))
        else:
            N = len(batch["qlen"])
            cnt = (np.asarray(batch["qlen"]).astype(np.int64) % 7) + 1
            eo = np.zeros(N + 1, np.int64); eo[1:] = np.cumsum(cnt)
            same = gr.shape[0] == N and bool(np.array_equal(gr[:, 0], np.arange(N))) and bool(np.array_equal(gr[:, 1], np.asarray(batch["qlen"]).astype(np.int32))) \
                and bool(np.array_equal(go, eo)) and bool(np.array_equal(gw, (np.repeat(np.arange(N, dtype=np.int64), cnt) * 16 + 1).astype(np.uint32)))
        nbytes = int(np.asarray(batch["seqs"]).size)
        out = {"pairs": int(len(batch["qlen"])), "ranks": world, "bounds": [int(b) for b in sh["bounds"]],
               "scatter_ms": round((t1 - t0) * 1e3, 2), "align_ms": round((t2 - t1) * 1e3, 2), "gather_ms": round((t3 - t2) * 1e3, 2),
               "round_trip_ms": round((t1 - t0 + t3 - t2) * 1e3, 2), "input_MB": round(nbytes / 1e6, 1), "result_MB": round((gr.nbytes + gw.nbytes) / 1e6, 1),
               "backend": dist.get_backend(), "gathered_identical_to_rank0_whole_batch": same,
               "what": "rank 0 scatters the batch (bsalign_amd/shard.py: bsa_shard_pack + grouped point-to-point sends), every rank %s, rank 0 gathers "
                       "records + CIGAR words in pair order" % ("aligns its shard" if align is not None else "fills stand-in records (no GPU)")}
    if own:
        dist.destroy_process_group()
    return out


def launch_check(args):
    rank, world, local, dist = init_ranks(args)
    if dist is not None:
        dist.barrier()
    import torch
    have_gpu = torch.cuda.is_available()
    dev = torch.device("cuda", local) if have_gpu else torch.device("cpu")
    ex = None
    if dist is not None and not args.no_exchange:
        # a small batch of synthetic pairs with ragged lengths through the exchange (stand-in records: no kernels run here)
        batch = None
        if rank == 0:
            rng = np.random.default_rng(SEED)
            N = 3000
            tl = rng.integers(200, 1200, N).astype(np.uint32); ql = (tl.astype(np.int64) + rng.integers(-50, 50, N)).astype(np.uint32)
            to = np.zeros(N, np.uint64); qo = np.zeros(N, np.uint64); acc = 0
            for k in range(N):
                to[k] = acc; acc += int(tl[k]); qo[k] = acc; acc += int(ql[k])
            batch = dict(seqs=rng.integers(0, 4, acc).astype(np.uint8), qoff=qo, qlen=ql, toff=to, tlen=tl)
        ex = exchange_selfcheck(dist, rank, world, dev, batch, 128)
    if rank == 0:
        emit({"launch_check": True, "n_gpus": world, "ranks_counted": world, "backend": dist.get_backend() if dist is not None else None, "exchange": ex})
    if dist is not None:
        dist.destroy_process_group()


def _cpu_sample(args, L, bw, sc, mode, npairs, first_pair, kind):
    """time the reference (oracle/_ref) or the own restatement on `npairs` synthetic pairs starting at `first_pair`,
    on the calling thread -> (band cells, seconds)"""
    import support as S
    import bsalign_amd as B
    stride = B.lib().bsa_synth_stride(L)
    seqs = np.zeros(2 * npairs * stride, dtype=np.uint8)
    qlen = np.zeros(npairs, dtype=np.uint32)
    B.lib().bsa_synth_pairs_host(SEED, first_pair, npairs, L, int(args.eps * 4294967296.0), seqs.ctypes.data_as(C.c_void_p), qlen.ctypes.data_as(C.c_void_p))
    tlen = np.full(npairs, L, dtype=np.uint32)
    toff = (np.arange(npairs, dtype=np.uint64) * np.uint64(stride))
    qoff = ((np.arange(npairs, dtype=np.uint64) + np.uint64(npairs)) * np.uint64(stride))
    cs = C.c_int64(0)
    if args.workload == "align8":
        cells = float(L) * bw * npairs if bw else float(L) * float(((qlen.astype(np.int64) + 15) // 16 * 16).sum())     # bandwidth 0 = the whole query (bsalign.h:3861)
        if kind == "reference":
            lib = S.ref()
            secs = lib.ref_align_batch_time(lib._ctx, S.ptr(seqs, S.u8p), S.ptr(qoff, S.u64p), S.ptr(qlen, S.u32p), S.ptr(toff, S.u64p),
                                            S.ptr(tlen, S.u32p), npairs, mode, bw, sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], C.byref(cs))
        else:
            m = S.score_matrix(sc[0], sc[1])
            secs = S.oracle().orc_align_batch_time(S.ptr(seqs, S.u8p), S.ptr(qoff, S.u64p), S.ptr(qlen, S.u32p), S.ptr(toff, S.u64p),
                                                   S.ptr(tlen, S.u32p), npairs, mode, bw, S.ptr(m, S.i8p), sc[2], sc[3], sc[4], sc[5], C.byref(cs))
    else:
        def bw_eff(ql, tl):                                    # bsalign.h:1055-1067
            qr = (int(ql) + 63) // 64 * 64
            if mode != B.MODE_GLOBAL:
                return qr
            b = (bw + 63) // 64 * 64
            if b == 0 or b > ql:
                b = qr
            if b < ql and b < (int(ql) + int(tl) - 1) // int(tl) + 1:
                b = ((int(ql) + int(tl) - 1) // int(tl) + 1 + 63) // 64 * 64
            return b
        cells = float(sum(int(tl) * bw_eff(ql, tl) for ql, tl in zip(qlen, tlen)))
        if kind == "reference":
            lib = S.ref()
            secs = lib.ref_edit_batch_time(lib._ctx, S.ptr(seqs, S.u8p), S.ptr(qoff, S.u64p), S.ptr(qlen, S.u32p), S.ptr(toff, S.u64p),
                                           S.ptr(tlen, S.u32p), npairs, mode, bw, C.byref(cs))
        else:
            secs = S.oracle().orc_edit_batch_time(S.ptr(seqs, S.u8p), S.ptr(qoff, S.u64p), S.ptr(qlen, S.u32p), S.ptr(toff, S.u64p),
                                                  S.ptr(tlen, S.u32p), npairs, mode, bw, C.byref(cs))
    return cells, secs


def physical_cores():
    """one logical CPU per physical core among the CPUs this process may run on (/proc/cpuinfo: physical id, core id)"""
    allowed = sorted(os.sched_getaffinity(0))
    core_of = {}
    try:
        cpu = phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("processor"):
                cpu = int(ln.split(":")[1]); phys = core = None
            elif ln.startswith("physical id"):
                phys = int(ln.split(":")[1])
            elif ln.startswith("core id"):
                core = int(ln.split(":")[1])
                core_of[cpu] = (phys, core)
    except OSError:
        pass
    seen, out = set(), []
    for c in allowed:
        key = core_of.get(c, ("cpu", c))
        if key not in seen:
            seen.add(key); out.append(c)
    return out


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), None = unlimited"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else max(1, int(float(q) / float(per) + 0.999))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, (q + per - 1) // per)
    except (OSError, ValueError):
        return None


def cpu_worker(args):
    """hidden mode of the all-core baseline: `--cpu-worker cpu,first_pair,npairs,t_go,L,bw,mode,kind` pins itself to one
    CPU, prepares its own slice of the synthetic pairs, waits for the common start time and prints what it measured"""
    import time
    f = args.cpu_worker.split(",")
    cpu, first, npairs, t_go, L, bw, mode, kind = int(f[0]), int(f[1]), int(f[2]), float(f[3]), int(f[4]), int(f[5]), int(f[6]), f[7]
    try:
        os.sched_setaffinity(0, {cpu})
    except OSError:
        pass
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sc = tuple(int(x) for x in args.scoring.split(","))
    _cpu_sample(args, L, bw, sc, mode, 8, first, kind)          # page in the code and the libraries
    while time.time() < t_go:
        pass
    cells, secs = _cpu_sample(args, L, bw, sc, mode, npairs, first, kind)
    print(json.dumps({"cells": cells, "secs": secs, "end": time.time()}))


def cpu_baseline(args, L, bw, sc, mode):
    """reference (oracle/_ref, kind 'reference') or own restatement (kind 'port') on the host: one process per physical
    core over disjoint slices of the synthetic pairs (aggregate = all cells / time from the common start to the last
    worker's end), and the single-core figure beside it.  Bounded sample, SURVEY.md section 8(d)."""
    import subprocess
    import time
    import support as S
    kind = "reference" if S.have_ref() else "port"
    if args.workload == "align8":
        npairs = args.cpu_pairs if args.cpu_pairs > 0 else (4000 if kind == "reference" else 1200)
        if args.cpu_pairs <= 0:
            # the sample is sized for bandwidth 128; wider bands (bandwidth 0 = the whole query) cost more per pair: keep it at 10 - 30 s
            eff = bw if bw else (L + 15) // 16 * 16
            npairs = max(16, int(npairs * min(1.0, 128.0 / eff)))
    else:
        npairs = args.cpu_pairs if args.cpu_pairs > 0 else (1500 if kind == "reference" else 60)
    what = ("the reference's SSE4.2 code (oracle/_ref)" if kind == "reference" else "own scalar C restatement (oracle/), not the reference")
    cells1, secs1 = _cpu_sample(args, L, bw, sc, mode, npairs, 0, kind)
    one = round(cells1 / secs1 / 1e9, 4)
    cpus = physical_cores()
    nphys = len(cpus)
    quota = cpu_quota()
    if quota is not None and quota < len(cpus):
        cpus = cpus[:: max(1, len(cpus) // quota)][:quota]      # spread over the sockets
    out = {"value": one, "unit": "GCUPS", "cores": 1, "kind": kind,
           "sample": "%d of the same synthetic pairs (L=%d, bw=%d), single thread, %s, %.1f s" % (npairs, L, bw, what, secs1)}
    if len(cpus) < 2:
        return out
    per = max(8, npairs // 2)                                  # pairs per core: about half of the single-core sample's time each
    t_go = time.time() + 8.0 + 0.06 * len(cpus)          # common start: every worker has loaded its libraries and made its pairs by then
    procs = []
    for k, cpu in enumerate(cpus):
        spec = "%d,%d,%d,%.3f,%d,%d,%d,%s" % (cpu, npairs + k * per, per, t_go, L, bw, mode, kind)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", spec, "--workload", args.workload, "--eps", str(args.eps), "--scoring", args.scoring]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=dict(os.environ, BSA_NO_TORCH_PRELOAD="1")))
    cells = 0.0; end = 0.0; ok = 0; slow = 0.0
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=600)
            r = json.loads(o.strip().splitlines()[-1])
            cells += r["cells"]; end = max(end, r["end"]); slow = max(slow, r["secs"]); ok += 1
        except Exception:
            pr.kill()
    if ok == 0 or end <= t_go:
        return out
    agg = round(cells / (end - t_go) / 1e9, 3)
    return {"value": agg, "unit": "GCUPS", "cores": ok, "kind": kind, "per_core": round(agg / ok, 4), "one_core_alone": one,
            "host_physical_cores": nphys, "cpu_quota": quota,
            "sample": "%d processes pinned to %d distinct physical cores (the host has %d; this container's CPU quota is %s), %d of the same "
                      "synthetic pairs each (L=%d, bw=%d), %s; all cells / time from the common start to the last process's end (%.1f s; "
                      "slowest process %.1f s in the DP); one core alone on %d pairs: %.4f GCUPS"
                      % (ok, len(cpus), nphys, "%d CPUs" % quota if quota else "unlimited", per, L, bw, what, end - t_go, slow, npairs, one)}


def poa_cpu_baseline(args, bw):
    """the reference's own align_rd_bspoacore (oracle/_ref) timed inside a real end_bspoa on synthetic reads, one core"""
    import support as S
    if not S.have_ref():
        return {"value": None, "unit": "GCUPS", "cores": 1, "kind": "reference", "sample": "oracle/_ref/libbsref.so absent: not measured"}
    import poa_support as P
    nreads, L = (args.cpu_pairs if args.cpu_pairs > 0 else 48), 6000
    reads = P.synth_reads(SEED & 0xFFFF, L, nreads, eps=(args.eps,))
    r = P.run_ref_poa(reads, 1, P.par(bandwidth=bw), record=False)
    return {"value": round(r["core_updates"] * bw / r["core_seconds"] / 1e9, 4), "unit": "GCUPS", "cores": 1, "kind": "reference",
            "sample": "reference end_bspoa on %d synthetic reads x %d bp (eps %.2f, default POA parameters, bandwidth %d): %d row updates + %d merges in "
                      "align_rd_bspoacore, %.2f s inside it, single thread (oracle/_ref)" % (nreads, L, args.eps, bw, r["core_updates"], r["core_merges"], r["core_seconds"])}


def main_poa_recorded(args):
    """`--workload poa --poa-source recorded`: the sweeps of REAL POA windows.  W windows of R synthetic reads each are run through the
    reference's end_bspoa (oracle/_ref) once, outside the timed region, and for every read the graph-form program the binding builds
    (include/bsalign_poa_adapter.h: nodes, in-edges, candidates) is recorded; the timed region runs all of them on the device the
    way the lock-step batcher does -- read r of all windows as one launch per band width: forward DP as a wavefront, best end cell,
    traceback (bsa_poa_graph_run).  Every program's best end cell is checked against what the reference's own sweep found.
    `--poa-source fixture` runs the same launches on the programs committed in tests/golden/poa_graph.npz (case 0, the default
    parameters; every window the same six reads) and needs no reference build: no cpu_baseline and no end-to-end leg then.
    Single GPU."""
    import torch
    import bsalign_amd as B
    import support as S
    import poa_support as P
    fixture = args.poa_source == "fixture"
    if not fixture and not S.have_ref():
        emit({"error": "oracle/_ref/libbsref.so absent: recorded POA programs need the reference build (--poa-source fixture runs the committed ones)"})
        return
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    nwin = args.pairs or 256
    ncores = len(physical_cores())
    q = cpu_quota()
    ncores = min(ncores, q) if q else ncores
    if fixture:
        # the programs the reference recorded into tests/golden/poa_graph.npz (case 0: default parameters), every window the same six reads
        cases = P.load_golden_graph()
        case = cases[max(0, min(args.poa_case, len(cases) - 1))]
        pp = case["par"]
        nreads, L = len(case["reads"]), max(rc["slen"] for rc in case["reads"])
        rec = [dict(recs=case["reads"]) for _ in range(nwin)]
        windows, t_ref, core_seconds, updates, merges = None, None, 0.0, 0.0, 0.0
    else:
        nreads, L = 12, (args.length or 1500)
        pp = P.par()
        windows = [P.synth_reads((SEED + 977 * w) & 0x7FFFFFFF, L, nreads, eps=(args.eps,)) for w in range(nwin)]
        _, t_ref = P.run_many(windows, 0, pp, threads=ncores)
        rec, _ = P.run_many(windows, 1, pp, threads=ncores, record=2)
        core_seconds = sum(w["core_seconds"] for w in rec)
        updates = sum(w["core_updates"] for w in rec)
        merges = sum(w["core_merges"] for w in rec)
    lib = B.lib()
    ctx = B.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    groups = {}
    for w, d in enumerate(rec):
        for r, rc in enumerate(d["recs"]):
            if rc["bandwidth"] <= 256 and len(rc["nodes"]) > 1:
                groups.setdefault((r, rc["bandwidth"]), []).append(rc)
    launches, cells, balg, nprog_total = [], 0.0, 0.0, 0
    for (r, bw), rcs in sorted(groups.items()):
        sp = B.SweepParams()
        sp.rows = B.RowsParams(pp["alnmode"], bw, pp["M"], pp["X"], pp["refbonus"], pp["O"], pp["E"], pp["Q"], pp["P"])
        sp.T = pp["T"]
        blk = lib.bsa_rows_block_bytes(bw, pp["O"], pp["E"], pp["Q"], pp["P"])
        progs = np.zeros(len(rcs), B.POA_PROG_DTYPE)
        n0 = e0 = c0 = q0 = v0 = 0
        for i, rc in enumerate(rcs):
            cap = 2 * (rc["slen"] + len(rc["nodes"])) + 64
            progs[i] = (n0, len(rc["nodes"]), e0, len(rc["edges"]), c0, len(rc["cands"]), rc["slen"], cap, q0, v0)
            n0 += len(rc["nodes"]); e0 += len(rc["edges"]); c0 += len(rc["cands"]); q0 += (rc["slen"] + 31) & ~15; v0 += cap
            nd = rc["nodes"]
            for tk in (nd["in0_tk"], nd["in1_tk"]):
                pres = (tk & 0x80000000) != 0
                nu = int((pres & ((tk & 0x40000000) == 0)).sum()); nm = int((pres & ((tk & 0x40000000) != 0)).sum())
                cells += float(nu) * bw; balg += (2.0 * nu + 3.0 * nm) * blk
                if fixture:
                    updates += nu; merges += nm
        qb = np.zeros(q0 + 64, np.uint8)
        for i, rc in enumerate(rcs):
            qb[int(progs[i]["query_off"]):int(progs[i]["query_off"]) + rc["slen"]] = rc["query"]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).to(dev)
        dv = dict(nodes=t(np.concatenate([rc["nodes"] for rc in rcs])), edges=t(np.concatenate([rc["edges"] for rc in rcs])),
                  cands=t(np.concatenate([rc["cands"] for rc in rcs])), progs=t(progs), q=t(qb),
                  res=torch.zeros(len(rcs) * 8, dtype=torch.int32, device=dev), steps=torch.zeros(v0 + 16, dtype=torch.int32, device=dev),
                  packed=torch.zeros(v0 + 16, dtype=torch.int32, device=dev), used=torch.zeros(2, dtype=torch.int64, device=dev))
        launches.append((sp, len(rcs), n0, max(rc["slen"] for rc in rcs), dv, rcs))
        nprog_total += len(rcs)

    def launch(sp, n, nn, msl, dv):
        rc = lib.bsa_poa_graph_run(ctx.h, dv["nodes"].data_ptr(), nn, dv["edges"].data_ptr(), dv["cands"].data_ptr(), dv["progs"].data_ptr(), n, dv["q"].data_ptr(),
                                   msl, C.byref(sp), dv["res"].data_ptr(), dv["steps"].data_ptr(), dv["packed"].data_ptr(), dv["used"].data_ptr(), None, None)
        assert rc == 0, "bsa_poa_graph_run failed: %d" % rc

    def step():
        for sp, n, nn, msl, dv, _ in launches:
            launch(sp, n, nn, msl, dv)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms_tot = 0.0
    for sp, n, nn, msl, dv, _ in launches:
        launch(sp, n, nn, msl, dv)
        kms_tot += ctx.last_kernel_ms()[0]
    ident, steps_total = True, 0
    for sp, n, nn, msl, dv, rcs in launches:
        res = dv["res"].cpu().numpy().view(B.POA_RESULT_DTYPE)
        for i, rc in enumerate(rcs):
            gi = int(rc["nodes"][int(res[i]["maxidx"])]["gnode"]) if res[i]["maxidx"] >= 0 else -1
            ident &= res[i]["status"] == 0 and (int(res[i]["maxscr"]), gi, int(res[i]["maxoff"])) == (rc["maxscr"], rc["maxidx"], rc["maxoff"])
            steps_total += int(res[i]["nevents"])
    # lock-step end to end: the reference's host code per window on host threads, every sweep + walk through the batcher
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    e2e = None
    try:
        if fixture:
            raise RuntimeError("not run: the end-to-end leg drives the reference's own host code (oracle/_ref)")
        if args.cpu_pairs < 0:
            raise RuntimeError("not run (--cpu-pairs -1: the timed launches only, as the profiler's counter passes want them)")
        from test_poa_batched_gpu import Batcher
        bt = Batcher(ctx, nwin)
        try:
            devres, t_dev = P.run_many(windows, 7, pp)
            st = bt.stats()
        finally:
            bt.close()
        base, _ = P.run_many(windows, 0, pp, threads=ncores)
        same = all(np.array_equal(a["cns"], b["cns"]) and a["msa"] == b["msa"] for a, b in zip(base, devres))
        e2e = {"windows_per_s": round(nwin / t_dev, 2), "seconds": round(t_dev, 3), "identical_to_reference": bool(same), "batches": st["batches"], "launches": st["launches"],
               "MB_up": round(st["bytes_up"] / 1e6, 1), "MB_down": round(st["bytes_down"] / 1e6, 1), "device_s": round(st["device_us"] / 1e6, 3),
               "note": "host side = the reference's own C (oracle/_ref) on one thread per window, one runnable thread per CPU; informational, not the measured value"}
    except Exception as ex:          # the checker is optional here
        e2e = {"error": str(ex)}
    nl = len(launches)
    achieved = (balg / nl) / (kms_tot / nl / 1e3) / 1e9 if kms_tot > 0 else 0.0
    shape = "poa-%s|n%d|reads%d|L%d" % (("fixture%d" % args.poa_case if args.poa_case else "fixture") if fixture else "recorded", nwin, nreads, L)
    pardesc = ("default POA parameters (overlap, bandwidth 128, 2-piece gaps)" if (not fixture or args.poa_case == 0) else
               "mode %d, bandwidth %d, M %d X %d O %d E %d Q %d P %d" % (pp["alnmode"], pp["bandwidth"], pp["M"], pp["X"], pp["O"], pp["E"], pp["Q"], pp["P"]))
    ent = counters_for(shape, "k_poa_wf")
    traffic = traffic_of(ent)
    if fixture:
        cpub = {"value": None, "unit": "GCUPS", "cores": 0, "kind": "reference", "sample": "not measured: oracle/_ref is absent and the fixture holds programs, not reads (the default "
                "source, recorded, times the reference's align_rd_bspoacore on the same windows)"}
    else:
        cpub = {"value": round(updates * 128 / (t_ref * 0 + core_seconds / ncores) / 1e9, 4) if core_seconds > 0 else None, "unit": "GCUPS", "cores": ncores, "kind": "reference",
                         "sample": "the reference's align_rd_bspoacore inside end_bspoa of the same %d windows on %d host threads (one window per thread at a time): %.2f s summed over threads "
                                   "= %.4f GCUPS per core; the whole end_bspoa of all windows: %.2f s = %.1f windows/s" % (nwin, ncores, core_seconds, updates * 128 / core_seconds / 1e9 if core_seconds > 0 else 0, t_ref, nwin / t_ref),
                         "end_bspoa_windows_per_s": round(nwin / t_ref, 2), "threads": ncores}
    line = {
        "metric": "GCUPS (giga band-cell updates / s), POA seq->graph DP (align_rd_bspoacore) + traceback (alignment2graph_bspoa) on " +
                  ("the committed fixture programs, tiled: every window the same reads" if fixture else "recorded programs"),
        "value": round(cells * args.steps / elapsed / 1e9, 3), "unit": "GCUPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32 (exact image of the reference's i8 differences)",
        "data": ("graph-form programs the reference's end_bspoa recorded into tests/golden/poa_graph.npz (case %d), every window the same %d reads" % (args.poa_case, nreads)) if fixture else
                "graph-form programs recorded from the reference's end_bspoa (oracle/_ref, outside the timed region) on synthetic reads, seed %d" % SEED,
        "config": {"workload": (("poa-fixture: %d POA windows (identical, tiled) x %d reads x %d bp, " % (nwin, nreads, L)) if fixture else
                                ("poa-recorded: %d POA windows x %d reads x %d bp (eps %.2f), " % (nwin, nreads, L, args.eps))) + pardesc +
                               ("; a step = every read's sweep and traceback of every window, read r of all windows per launch (%d launches, %d programs, %.0f row updates + %.0f merges, "
                                "%d traceback steps)" % (nl, nprog_total, updates, merges, steps_total)),
                   "shape": shape, "windows": nwin, "pairs_per_gpu": nwin, "bandwidth": int(pp["bandwidth"]), "reads": nreads, "length": L, "sweep_windows_per_s": round(nwin * args.steps / elapsed, 1)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "kernel": "k_poa_wf<%d, %d> (row-at-a-time forward pass, plain nodes in copies of their own + traceback in tiles)" % (
                         2 if (pp["Q"] or pp["P"]) else 1 if pp["O"] else 0, 1 if int(pp["bandwidth"]) <= 64 else 2 if int(pp["bandwidth"]) <= 128 else 4), "kernel_ms_avg": round(kms_tot / nl, 3), "launches_per_step": nl, "algorithmic_bytes_per_launch": round(balg / nl, 1),
                     "issue": issue_fractions(ent, kms_tot / nl),
                     "traffic_from": (ent or {}).get("source") if traffic else None,
                     "note": "not HBM-bound: one wave per read, a graph node per trip (about 180 instructions); a lone wave is bound by its own dependent latencies, thousands of "
                             "windows in flight by VALU issue (HISTORY section 4b); about 6 KB of LDS per read, five waves per SIMD"},
        "checks": {"best_end_cell_identical_all_programs": bool(ident), "programs": nprog_total},
        "cpu_baseline": cpub,
        "lockstep_end_to_end": e2e,
    }
    emit(line)
    ctx.close()


def main_poa(args):
    """C4-shaped workload: the per-read sweep (align_rd_bspoacore) of many POA windows side by side, one program per window"""
    if args.poa_source == "auto":
        import support as S
        args.poa_source = "synthetic" if int(os.environ.get("WORLD_SIZE", "1")) != 1 else "recorded" if S.have_ref() else "fixture"
    if args.poa_source in ("recorded", "fixture"):
        return main_poa_recorded(args)
    import torch
    import bsalign_amd as B
    from bsalign_amd import poa_synth as PS
    rank, world, local, dist = init_ranks(args)
    dev = torch.device("cuda", local)
    nwin = args.pairs or 16384                                 # 4 programs per wave: 4096 waves = 4 per SIMD
    npos = args.length or 10000
    bw = args.bw or 128
    K = 8                                                    # distinct programs, windows cycle through them
    pp = dict(alnmode=1, M=2, X=-6, O=-3, E=-2, Q=-8, P=-1, T=20, refbonus=1)     # DEFAULT_BSPOA_PAR, bspoa.h:79-81
    lib = B.lib()
    sp = B.SweepParams()
    sp.rows = B.RowsParams(pp["alnmode"], bw, pp["M"], pp["X"], pp["refbonus"], pp["O"], pp["E"], pp["Q"], pp["P"])
    sp.T = pp["T"]
    blk = lib.bsa_rows_block_bytes(bw, pp["O"], pp["E"], pp["Q"], pp["P"])
    rng = np.random.default_rng(SEED + rank)
    tasks, first, ntask, nblk, nupd, nmrg, queries, qoff, qlen = [], [], [], [], [], [], [], [], []
    tacc = qacc = 0
    for k in range(K):
        t, nb, nu, nm, slen = PS.make_program(SEED + 1000 * rank + k, npos, bw)
        t["query"] = k
        tasks.append(t)
        first.append(tacc)
        ntask.append(len(t))
        tacc += len(t)
        nblk.append(nb)
        nupd.append(nu)
        nmrg.append(nm)
        queries.append(rng.integers(0, 4, size=slen).astype(np.uint8))
        qoff.append(qacc)
        qlen.append(slen)
        qacc += (slen + 63) & ~63
    qblob = np.zeros(qacc + 64, dtype=np.uint8)
    for k in range(K):
        qblob[qoff[k]:qoff[k] + qlen[k]] = queries[k]
    tasks = np.concatenate(tasks)
    progs = np.zeros(nwin, dtype=B.SWEEP_PROG_DTYPE)
    kk = np.arange(nwin) % K
    progs["first_task"] = np.array(first, dtype=np.uint32)[kk]
    progs["ntasks"] = np.array(ntask, dtype=np.uint32)[kk]
    blocks = np.array(nblk, dtype=np.int64)[kk]
    progs["first_block"] = np.concatenate([[0], np.cumsum(blocks)[:-1]]).astype(np.uint32)
    total_blocks = int(blocks.sum())
    updates = float(np.array(nupd, dtype=np.int64)[kk].sum())
    merges = float(np.array(nmrg, dtype=np.int64)[kk].sum())
    cells = updates * bw
    ctx = B.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_rows = torch.empty(total_blocks * blk, dtype=torch.uint8, device=dev)
    d_tasks = torch.from_numpy(tasks.view(np.uint8)).to(dev)
    d_progs = torch.from_numpy(progs.view(np.uint8)).to(dev)
    d_q = torch.from_numpy(qblob).to(dev)
    d_qoff = torch.from_numpy(np.array(qoff, dtype=np.int64)).to(dev)
    d_qlen = torch.from_numpy(np.array(qlen, dtype=np.int32)).to(dev)
    d_res = torch.zeros(nwin * 4, dtype=torch.int32, device=dev)

    def step():
        rc = lib.bsa_sweep_run(ctx.h, d_rows.data_ptr(), d_tasks.data_ptr(), d_progs.data_ptr(), nwin, d_q.data_ptr(), d_qoff.data_ptr(),
                               d_qlen.data_ptr(), C.byref(sp), d_res.data_ptr())
        assert rc == 0, "bsa_sweep_run failed: %d" % rc

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        barrier()
    kms, klaunch, _ = ctx.last_kernel_ms()
    res = d_res.cpu().numpy().view(B.SWEEP_RESULT_DTYPE)
    if rank == 0:
        import poa_support as P
        # parity spot check (outside the timed region): the K distinct programs vs the oracle, rows and results
        ident = True
        rows_host = d_rows[: int(blocks[:K].sum()) * blk].cpu().numpy()
        pw = S_oracle_piecewise(pp, bw)
        used = bw * (pw + 1) + 68
        for k in range(K):
            t = tasks[first[k]:first[k] + ntask[k]].copy()
            t["query"] = 0
            orows, ores = P.oracle_sweep(t, np.array([(0, ntask[k], 0, 0)], dtype=P.PROG_DTYPE), queries[k], np.zeros(1, np.uint64),
                                         np.array([qlen[k]], np.uint32), dict(pp), bw, nblk[k], pw)
            b0 = int(progs[k]["first_block"])
            mine = rows_host[b0 * blk:(b0 + nblk[k]) * blk].reshape(nblk[k], blk)[1:, :used]
            ident &= bool(np.array_equal(mine, orows.reshape(nblk[k], blk)[1:, :used]))
            ident &= (int(res[k]["maxscr"]), int(res[k]["maxidx"]), int(res[k]["maxoff"])) == (int(ores[0]["maxscr"]), int(ores[0]["maxidx"]), int(ores[0]["maxoff"]))
        # algorithmic bytes: one row block read + one written per row update; a merge reads two and writes one
        balg = (2.0 * updates + 3.0 * merges) * blk
        achieved = balg / (kms / 1e3) / 1e9 if kms > 0 else 0.0
        shape = "poa|n%d|pos%d|bw%d" % (nwin, npos, bw)
        ent = counters_for(shape, "k_sweep")
        traffic = traffic_of(ent)
        line = {
            "metric": "GCUPS (giga band-cell updates / s), POA seq->graph DP sweep (align_rd_bspoacore)",
            "value": round(cells * args.steps * world / elapsed / 1e9, 3), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8",
            "data": "synthetic sweep programs (bsalign_amd/poa_synth.py: op mix of the reference's real programs), seed %d" % SEED,
            "config": {"workload": "poa: %d POA windows/GPU, one read-vs-graph sweep each over %d graph positions (%.0f row updates + %.0f merges per window), "
                                   "overlap mode, bandwidth %d, DEFAULT_BSPOA_PAR scoring (2-piece gaps)" % (nwin, npos, updates / nwin, merges / nwin, bw),
                       "shape": shape, "windows_per_gpu": nwin, "positions": npos, "bandwidth": bw,
                       "parallelism": "windows sharded across %d GPU(s), no data-path collective" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "kernel": "k_sweep", "kernel_ms_avg": round(kms, 3), "launches_per_step": klaunch,
                         "algorithmic_bytes_per_launch": round(balg, 1)},
            "checks": {"oracle_identical_first8_programs": ident},
        }
        if world == 1 and args.cpu_pairs >= 0:
            line["cpu_baseline"] = poa_cpu_baseline(args, bw)
        emit(line)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def S_oracle_piecewise(pp, bw):
    import support as S
    return int(S.oracle().orc_get_piecewise(pp["O"], pp["E"], pp["Q"], pp["P"], bw))


_REAL_STDOUT = None


def quiet_stdout():
    """ONE JSON line on stdout is the contract, and libraries talk there (RCCL prints a version banner on the first communicator): from here on fd 1 is
    stderr, the JSON line goes out through emit() on the descriptor kept aside"""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    sys.stdout.flush()
    data = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.buffer.write(data); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def main():
    args = parse()
    if args.cpu_worker:
        return cpu_worker(args)
    ensure_ranks(args)
    quiet_stdout()
    if args.launch_check:
        return launch_check(args)
    if args.workload == "poa":
        return main_poa(args)
    import torch
    import bsalign_amd as B
    rank, world, local, dist = init_ranks(args)
    dev = torch.device("cuda", local)
    sc = tuple(int(x) for x in args.scoring.split(","))
    mode = {"global": B.MODE_GLOBAL, "overlap": B.MODE_OVERLAP, "extend": B.MODE_EXTEND}[args.mode]
    if args.workload == "align8":
        n = args.pairs or 100000
        L = args.length or 10000
        bw = args.bw or 128
    else:
        n = args.pairs or 32768          # 100 kbp pairs at bandwidth 256: 64 bytes a row, 210 GB of row planes -- the batch one MI355X holds (16384 pairs are two waves per SIMD: 4.8 k GCUPS against 6.2 k)
        L = args.length or 100000
        bw = args.bw or 256
    if bw < 0:
        bw = 0

    ctx = B.Context(local, int(args.workspace_gb * (1 << 30)))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = B.lib()
    stride = lib.bsa_synth_stride(L)
    d_seqs = torch.empty(2 * n * stride, dtype=torch.uint8, device=dev)
    d_qlen = torch.empty(n, dtype=torch.int32, device=dev)
    rc = lib.bsa_synth_pairs_dev(ctx.h, SEED, rank * n, n, L, int(args.eps * 4294967296.0), C.c_void_p(d_seqs.data_ptr()), C.c_void_p(d_qlen.data_ptr()))
    assert rc == 0, "synthetic generator failed"
    torch.cuda.synchronize()
    qlen = d_qlen.cpu().numpy().astype(np.uint32)
    tlen = np.full(n, L, dtype=np.uint32)
    toff = np.arange(n, dtype=np.uint64) * np.uint64(stride)
    qoff = (np.arange(n, dtype=np.uint64) + np.uint64(n)) * np.uint64(stride)

    par = B.make_params(mode, bw, *sc) if args.workload == "align8" else None

    def make_plan(qo, ql, to, tl):
        return B.AlignPlan(ctx, qo, ql, to, tl, par) if args.workload == "align8" else B.EditPlan(ctx, qo, ql, to, tl, mode, bw)

    tp0 = time.perf_counter()
    plan = make_plan(qoff, qlen, toff, tlen)
    plan_ms = (time.perf_counter() - tp0) * 1e3          # host planning of the batch (slot layout, chunking, kernel choice): outside the timed region, reported in config.plan_ms
    cells = plan.cells()
    cig_cap = int(n) * max(L // 4, 64)
    d_out = torch.zeros(n * 10, dtype=torch.int32, device=dev)
    d_cig = torch.empty(cig_cap, dtype=torch.int32, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)

    def step():
        plan.run(d_seqs, d_out, d_cig, d_off, d_st)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    per_rank = [round(cells * args.steps / elapsed / 1e9, 3)]
    if dist is not None:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, mine)
        allc = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allc, torch.tensor([float(cells)], dtype=torch.float64, device=dev))
        per_rank = [round(float(c) * args.steps / float(t) / 1e9, 3) for c, t in zip(allc.cpu().tolist(), allt.cpu().tolist())]
        tt = mine.clone()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        barrier()
    kms, klaunch, kcells = ctx.last_kernel_ms()
    tms, tlaunch = ctx.last_trace_ms()
    fwd_name, trace_name = ctx.last_kernel_names()

    out = d_out.cpu().numpy().reshape(n, 10)
    off = d_off.cpu().numpy()
    status = d_st.cpu().numpy()
    ncig = int(off[n])
    nbad = int((status != 0).sum())

    # ---- outside the timed region: what a caller with HOST buffers pays (PCIe-inclusive, never `value`), and the shard exchange on this very batch
    pcie_ms = None
    pcie_first_ms = None
    exchange = None
    host = None
    if not args.no_exchange and args.cpu_pairs >= 0:
        plan.close()
        plan = None
        if rank == 0:
            host = dict(seqs=d_seqs.cpu().numpy(), qoff=qoff, qlen=qlen, toff=toff, tlen=tlen)
            cig_all = d_cig[:ncig].cpu().numpy().view(np.uint32)
            # one host-pointer call of the whole batch: plan + upload + the same kernels + download
            h_out = np.zeros((n, 10), np.int32); h_cig = np.zeros(cig_cap, np.uint32); h_off = np.zeros(n + 1, np.uint64); h_st = np.zeros(n, np.uint32)
            fn = lib.bsa_align_batch if args.workload == "align8" else lib.bsa_edit_batch
            hp = par if args.workload == "align8" else B.EditParams(mode, bw)
            # twice on the SAME caller buffers: the first call also pays what the operating system and the driver charge for memory they have not seen
            # (the output arrays' pages are faulted in by the download: 10 instead of 55 GB/s on this box; the input pages are pinned for the first
            # time: 20 - 34 instead of 56 GB/s, tools/pcie_probe.hip), the second is what a caller that reuses its buffers pays per batch
            pcie_first_ms = None
            for rep in range(2):
                tq0 = time.perf_counter()
                rcb = fn(ctx.h, host["seqs"].ctypes.data_as(C.POINTER(C.c_uint8)), host["seqs"].size, qoff.ctypes.data_as(C.POINTER(C.c_uint64)), qlen.ctypes.data_as(C.POINTER(C.c_uint32)),
                         toff.ctypes.data_as(C.POINTER(C.c_uint64)), tlen.ctypes.data_as(C.POINTER(C.c_uint32)), n, C.byref(hp), h_out.ctypes.data_as(C.c_void_p),
                         h_cig.ctypes.data_as(C.POINTER(C.c_uint32)), cig_cap, h_off.ctypes.data_as(C.POINTER(C.c_uint64)), h_st.ctypes.data_as(C.POINTER(C.c_uint32)))
                pcie_ms = (time.perf_counter() - tq0) * 1e3 if rcb == 0 else None
                if rep == 0:
                    pcie_first_ms = pcie_ms
                if rcb != 0:
                    break
            pcie_same = rcb == 0 and bool(np.array_equal(h_out, out)) and bool(np.array_equal(h_off.astype(np.int64), off)) and bool(np.array_equal(h_cig[:ncig], cig_all))
            del h_cig

    if rank == 0:
        import support as S
        # parity spot check (outside the timed region): first pairs of the batch vs the oracle
        ident = True
        cig_host = d_cig[: int(off[8])].cpu().numpy().view(np.uint32) if n >= 8 else None
        for k in range(min(8, n)):
            q, t = S.synth_pair(k, L, err_q32=int(args.eps * 4294967296.0))
            if args.workload == "align8":
                res, cig, _ = S.oracle_align(q, t, mode, bw, *sc)
            else:
                res, cig, _ = S.oracle_edit(q, t, mode, bw)
            ident &= bool(np.array_equal(out[k], res)) and bool(np.array_equal(cig_host[int(off[k]):int(off[k + 1])], cig))
        # algorithmic bytes: 4-bit (8-bit path) / 2-bit (edit) traceback code per band cell +
        # sequences at 1 B/base + result struct + CIGAR words
        # (two-piece gaps: 8 bits per cell, SURVEY.md 8(d) / bsalign.h:47-54)
        two_piece = args.workload == "align8" and sc[4] < sc[2] and sc[5] > sc[3] and sc[4] + sc[5] < sc[2] + sc[3] and (sc[2] - sc[4]) // (sc[3] - sc[5]) < max(bw, 16)
        per_cell = (1.0 if two_piece else 0.5) if args.workload == "align8" else 0.25
        balg = per_cell * cells + float(qlen.sum()) + float(tlen.sum()) + 40.0 * n + 4.0 * ncig
        # the roofline line is about the kernel that takes most of the step: the forward DP or the traceback, whichever
        # the library's own HIP events (on the streams the kernels run on) say is longer per step
        dom_trace = tms * tlaunch > kms * klaunch
        dms, dlaunch, dname = (tms, tlaunch, trace_name) if dom_trace else (kms, klaunch, fwd_name)
        achieved = (balg / max(dlaunch, 1)) / (dms / 1e3) / 1e9 if dms > 0 else 0.0
        shape = "%s|n%d|L%d|%s|bw%d|%s" % (args.workload, n, L, args.mode, bw, args.scoring)
        ent = counters_for(shape, dname)
        traffic = traffic_of(ent)
        line = {
            "metric": "GCUPS (giga band-cell updates / s), %s" % ("8-bit banded striped global alignment" if args.workload == "align8" else "2-bit striped edit alignment"),
            "value": round(cells * args.steps * world / elapsed / 1e9, 3),
            "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i8" if args.workload == "align8" else "u64-bitplanes",
            "data": "synthetic (splitmix64 pairs, eps=%.2f, sub:ins:del=23:31:46, seed %d)" % (args.eps, SEED),
            "config": {"workload": "%s: %d pairs/GPU x %d bp, mode %s, bandwidth %d, scoring M,X,O,E,Q,P=%s" % (args.workload, n, L, args.mode, bw, args.scoring),
                       "shape": shape, "pairs_per_gpu": n, "length": L, "bandwidth": bw, "parallelism": "pairs sharded across %d GPU(s), no data-path collective" % world,
                       "timed": "stage + forward + traceback + CIGAR compaction on device-resident inputs; plan creation (host planning, slot layout) and PCIe are outside the timed region",
                       "plan_ms": round(plan_ms, 2),
                       "pcie_inclusive_ms": round(pcie_ms, 2) if pcie_ms is not None else None,
                       "pcie_inclusive_first_call_ms": round(pcie_first_ms, 2) if pcie_first_ms is not None else None,
                       "pcie_inclusive_what": "ONE host-pointer call (bsa_%s_batch) of the same batch on rank 0: plan + upload of the sequences from pageable memory + the same kernels + download of records and CIGAR words; "
                                              "the second of two calls on the same caller buffers (`pcie_inclusive_first_call_ms`: the first, whose output pages the download faults in and whose input pages the driver pins for the first time); "
                                              "never `value`" % ("align" if args.workload == "align8" else "edit"),
                       "per_rank_gcups": per_rank, "ranks_counted": world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": dname, "kernel_ms_avg": round(dms, 3), "launches_per_step": dlaunch,
                         "algorithmic_bytes_per_launch": round(balg / max(dlaunch, 1), 1),
                         "kernel_gcups": round(kcells / max(dlaunch, 1) / (dms / 1e3) / 1e9, 2) if dms > 0 else None,
                         "issue": issue_fractions(ent, dms),
                         "traffic_from": (ent or {}).get("source") if traffic else None,
                         "other_kernel": {"kernel": fwd_name if dom_trace else trace_name, "kernel_ms_avg": round(kms if dom_trace else tms, 3),
                                          "launches_per_step": klaunch if dom_trace else tlaunch}},
            "checks": {"pairs_flagged": nbad, "cigar_words": ncig, "oracle_identical_first8": ident},
        }
        if world == 1 and args.cpu_pairs >= 0:
            line["cpu_baseline"] = cpu_baseline(args, L, bw, sc, mode)
    # ---- the shard exchange on this very batch, after the line is complete: a fault or a hang of the exchange (RCCL at more than one rank has only
    # ever run here over gloo / shared memory: no multi-GPU node) must not cost the run its measured line
    if not args.no_exchange and args.cpu_pairs >= 0:
        import threading
        finished = threading.Event()
        limit = float(os.environ.get("BSA_BENCH_EXCHANGE_TIMEOUT", "300"))

        def watchdog():
            if not finished.wait(limit):
                if rank == 0:
                    line["exchange"] = {"error": "the shard exchange did not finish within %d s; the line is printed without it" % int(limit)}
                    line["exchange_ok"] = False                # top level, for the driver: the measured numbers above stand, the N-rank exchange did NOT work
                    emit(line)
                os._exit(0 if rank == 0 else 3)               # (a hung collective cannot be torn down in order; the other ranks leave with a failure code)

        if world > 1:
            threading.Thread(target=watchdog, daemon=True).start()

        def align_shard(sh):
            m = len(sh["qlen"])
            o = torch.zeros(max(m, 1) * 10, dtype=torch.int32, device=dev)
            f = torch.zeros(m + 1, dtype=torch.int64, device=dev)
            stt = torch.zeros(max(m, 1), dtype=torch.int32, device=dev)
            if m:
                pl = make_plan(sh["qoff"], sh["qlen"], sh["toff"], sh["tlen"])
                pl.run(sh["seqs"], o, d_cig, f, stt)
                ctx.sync()
                pl.close()
            return o.view(-1, 10)[:m], d_cig, f

        try:
            exchange = exchange_selfcheck(dist, rank, world, dev, host, bw if bw else 128, align=align_shard,
                                          expect=(out, cig_all, off) if rank == 0 else None)
            if rank == 0:
                exchange["host_pointer_call_identical"] = pcie_same
        except Exception as ex:          # (reported in the line, not raised)
            exchange = {"error": "%s: %s" % (type(ex).__name__, ex)} if rank == 0 else None
        finished.set()
        host = None
        if rank == 0 and exchange is not None:
            line["exchange"] = exchange
            line["exchange_ok"] = bool(exchange.get("gathered_identical_to_rank0_whole_batch")) and "error" not in exchange
    if plan is not None:
        plan.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        default_run = (world == 1 and args.workload == "align8" and not args.pairs and not args.length and not args.bw and args.mode == "global"
                       and args.scoring == "2,-6,-3,-2,0,0" and args.cpu_pairs >= 0 and not args.no_secondary and not os.environ.get("BSA_BENCH_NO_SECONDARY"))
        if default_run:
            # the other two configurations BASELINE.json names (C3, C4-shaped windows) in the same run, under the same clock: the device is free again here
            torch.cuda.empty_cache()
            line["secondary"] = secondary_lines(args)
        emit(line)


def secondary_lines(args):
    """`bench.py` with no workload flags also measures the edit path (C3: 32768 x 100 kbp, bandwidth 256), the POA's sweep + walk (4096 windows x 12 reads x
    1.5 kbp: programs recorded from the reference's end_bspoa where oracle/_ref exists, else the committed fixture programs) and whole-query bands with scores
    outside the static guard (2048 x 10 kbp, 10,-30,-20,-10) -- each as a run of this very script
    with its own steps, roofline and cpu_baseline, its JSON line embedded under "secondary"."""
    import subprocess
    out = []
    # (third: `bsalign align -W 0` with scores outside the static exactness guard -- the shape that ran below the CPU until round 6 -- on the checked systolic kernel)
    for extra in (["--workload", "edit", "--steps", "3", "--warmup", "1", "--no-exchange"],
                  ["--workload", "poa", "--pairs", "4096", "--steps", "5", "--warmup", "1"],
                  ["--pairs", "2048", "--length", "10000", "--bw", "-1", "--scoring", "10,-30,-20,-10,0,0", "--steps", "3", "--warmup", "1", "--no-exchange"]):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra, capture_output=True, text=True, timeout=900)
            ln = [x for x in r.stdout.splitlines() if x.startswith("{")]
            d = json.loads(ln[-1]) if ln else {"error": "no JSON line", "rc": r.returncode, "stderr_tail": r.stderr[-400:]}
        except Exception as ex:
            d = {"error": str(ex)}
        d["argv"] = " ".join(extra)
        d["wall_s"] = round(time.perf_counter() - t0, 1)
        out.append(d)
    return out


if __name__ == "__main__":
    main()
