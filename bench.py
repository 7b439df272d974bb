#!/usr/bin/env python3
"""bench.py -- decompress throughput of the Sprintz hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Headline workload (BASELINE.json configs[1], SURVEY.md 8d "cfg2"): uint16 row-major, 8
variables, FIRE predictor + bit-pack + RLE (sprintz_*_xff_16b), 10 KB chunks
(5120 elements), 131072 chunks (1.34 GB raw) PER GPU -- weak scaling, rank r
owns its own chunk range, no data-path collective; one all-gather of
compressed byte counts builds the global container layout.
One step = one batched decompress of the rank's whole batch, compressed
streams + offsets table resident in HBM, output written to HBM.
value = decompressed bytes of all ranks / max-over-ranks wall time (MB/s, 1e6).

The same JSON line carries `per_config`: every other BASELINE.json configuration (cfg1, cfg3 at
1 KB and 10 KB chunks, cfg4 = cfg2 + the Huff0 wire-format stage at 10 000 / 80 000 / 800 000
chunks, cfg5 = 1 M x 32 column-major), each with its own timing, ratio, roofline and a CPU
baseline of the compiled reference on this host.  cfg4 and cfg5 are FIXED-size workloads: with
--gpus N they are strong-scaled (chunks split over the ranks), the regime BASELINE.json states.

Synthetic data: SURVEY.md 8d's generator (tools/synth.py == oracle/synth.c: splitmix64 seeded per
chunk), per-column wrapping random walk with steps uniform in [-8, 8] (G1) unless --data says otherwise.
"""
import argparse
import ctypes as C
import json
import os
import platform
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
CPU_SAMPLE_BYTES = 1342177280   # raw bytes of the all-core CPU legs' sample (the headline batch of one GPU: 131 072 x 10 KB)
METRIC = "decompress MB/s (and ratio) uint16 rowmajor 8-col, 1/2/4/8 MI355X vs CPU ref"
# the decode kernel each configuration's batch runs on (names as rocprofv3 prints them: profiles/*_kernel_stats.csv)
DECODE_KERNELS = {"cfg1": "decode_uni_kernel<8, false, 1, 0>", "cfg3_1k": "verbatim_decode_kernel", "cfg3_10k": "decode_row_kernel<8>",
                  "cfg4": "decode_fast_kernel<16, true, 8, 1, true, 0, false, 0>"}
# ... and the kernels its compress call runs (round 6: large delta batches of the general layout take the block-parallel encoder, csrc/encode_blk.h)
ENCODE_KERNELS = {"cfg1": "encode_uni_kernel<8, false, 1> + scan_* + compact_copy_kernel", "cfg3_1k": "verbatim_dense_kernel",
                  "cfg3_10k": "encode_blk_kernel<8> + scan_* + compact_copy_kernel", "cfg4": "encode_wide_kernel<16, true, true, false, 4, false> (container inside the launch)"}
ALL_CONFIGS = ["cfg1", "cfg3_1k", "cfg3_10k", "cfg4_10000", "cfg4_80000", "cfg4_800000", "cfg5", "cfg5_8m"]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--nchunks", type=int, default=131072, help="chunks per GPU (headline)")
    p.add_argument("--data", default="walk8", choices=["walk8", "walk300", "uniform", "walkflat"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-verify", action="store_true", help="timing ablations only")
    p.add_argument("--cpu-seconds", type=float, default=6.0, help="target CPU work per headline baseline leg")
    p.add_argument("--ramp-ms", type=float, default=300.0,
                   help="untimed setup: keep the GPU busy with the same launches this long so that clocks settle "
                        "(0.51 ms/step cold vs 0.46 ms/step sustained was measured on MI355X)")
    p.add_argument("--configs", default="all", help="per_config entries to run: 'all', 'none' or a comma list of " + ",".join(ALL_CONFIGS))
    p.add_argument("--only", default="", help="profiling aid: run ONLY this per_config entry (no headline), print its JSON")
    p.add_argument("--config-reps", type=int, default=20)
    p.add_argument("--no-sweep", action="store_true", help="skip data_sweep (the headline shape on uniform / walk300 / walkflat data)")
    p.add_argument("--no-extras", action="store_true", help="skip the Huffman / query / latency extras of the headline batch")
    p.add_argument("--dry-launch", action="store_true",
                   help="launch check only: bring the N ranks up (self-launching them if need be), all-gather the ranks, print one JSON "
                        "line; needs no GPU (gloo without one)")
    p.add_argument("--strict-rccl", action="store_true",
                   help="with --gpus N > 1: abort before any timing if the layout exchange is not the library's own ncclAllGather "
                        "(csrc/comm.cpp behind the C-ABI); the default reports the fall-back to torch.distributed loudly on stderr and on the line")
    p.add_argument("--rccl", action="store_true",
                   help="with --dry-launch: also run the library's own layout gather (comm.cpp, ncclAllGather behind the C-ABI) over the N "
                        "ranks and check what every rank received -- seconds on a multi-GPU box, needs one GPU per rank")
    return p.parse_args()


DATA_KINDS = {"walk8": ("walk", 8), "walk300": ("walk", 300), "uniform": ("uniform", 0), "walkflat": ("walkflat", 8)}


def host_description():
    model, flags = "unknown", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            if line.startswith("flags") and not flags:
                flags = line.split(":", 1)[1]
    except OSError:
        pass
    fl = set(flags.split())
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "machine": platform.machine(), "cgroup_cpu_quota_cores": cgroup_cpu_limit()[0],
            "isa": {k: (k in fl) for k in ("avx2", "bmi2", "bmi1", "abm", "avx512f")},
            "reference_build": "g++ -O3 -mavx2 -mbmi2 -mbmi -mlzcnt (oracle/Makefile ref): the reference REQUIRES AVX2+BMI2 (sprintz_delta.h:21)"}


class Timer:
    """average duration of fn() over reps launches, HIP events on the stream the library launches on
    (torch's current stream -- the library is handed exactly that stream)"""

    def __init__(self, torch, default_ramp_ms=0.0):
        self.torch = torch
        self.default_ramp_ms = default_ramp_ms

    def __call__(self, fn, reps, warm=2, ramp_ms=None):
        """ramp_ms: run fn back to back for that long first -- the part clocks up under load, and a 5 ms timing that starts right after a host-side
        gap (a .item(), a verification) runs its first milliseconds at the idle clock: the same decode launch measured 0.267 ms cold and 0.225 ms
        after a leg that kept the GPU busy, same process (round 6).  The headline has had its own --ramp-ms since round 3."""
        t = self.torch
        if ramp_ms is None:
            ramp_ms = self.default_ramp_ms
        for _ in range(warm):
            fn()
        t.cuda.synchronize()
        if ramp_ms > 0:
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < ramp_ms:
                for _ in range(10):
                    fn()
                t.cuda.synchronize()
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        t.cuda.synchronize()
        return e0.elapsed_time(e1) / reps


# ------------------------------------------------------------------------------------------ CPU side
def cpu_libs():
    from tests.harness import ORACLE_SO, REF_SO
    if os.path.exists(REF_SO):
        return C.CDLL(REF_SO), "ref_decompress_chunks", "ref_decompress", "reference"
    if os.path.exists(ORACLE_SO):
        return C.CDLL(ORACLE_SO), "oracle_decompress_chunks", "oracle_decompress", "port"
    return None, None, None, None


def cgroup_cpu_limit():
    """CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable;
    plus the throttle counter, so that a run can show whether the all-core leg hit the limit"""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / per
        except (OSError, ValueError):
            pass
    throttled = None
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(f):
                if line.startswith("nr_throttled"):
                    throttled = int(line.split()[1])
            break
        except OSError:
            pass
    return quota, throttled


def host_topology():
    """-> (logical CPUs this process may run on, one logical CPU per PHYSICAL core among them)"""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    seen, firsts = set(), []
    for c in avail:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            firsts.append(c)
    return avail, firsts


def time_cpu_mt(call, n_one, n_all, raw_per_chunk, target_s):
    """call(nchunks, nthreads, reps, cpus) -> seconds per pass, all in C (oracle/mt_bench.c: pthreads, one chunk range per
    thread, pinned, sustained over `reps` passes).  Legs: 1 thread on the first n_one chunks; one thread per PHYSICAL core
    and one per LOGICAL CPU on the first n_all chunks.  -> dict of MB/s figures and what ran"""
    avail, firsts = host_topology()
    quota, thr0 = cgroup_cpu_limit()
    t = call(n_one, 1, 1, None)                                  # also faults the output pages in
    reps1 = max(1, min(50, int(target_s / max(t, 1e-6))))
    best1 = min(call(n_one, 1, reps1, None) for _ in range(2))
    r = {"value_1thread": round(n_one * raw_per_chunk / best1 / 1e6, 1), "physical_cores": len(firsts), "logical_cpus": len(avail)}
    # a container with a CPU quota (cgroup cpu.max) gets that many CPUs' worth of time however many threads it starts: more threads
    # than the quota only run in bursts between throttle periods (a burst shorter than one period even reads ABOVE the quota's rate).
    # The all-core leg therefore uses min(physical cores, quota) threads, and says so.
    limit = len(firsts) if quota is None else max(1, min(len(firsts), int(quota)))
    r["threads_limit"] = ("physical cores" if limit == len(firsts) else
                          f"cgroup cpu quota of {quota:g} CPUs (the host has {len(firsts)} physical cores)")
    legs = {}
    for name, cpus in (("physical", firsts[:limit]), ("logical", avail)):
        if name == "logical" and (len(avail) == len(firsts) or limit < len(firsts)):
            continue
        nt = min(len(cpus), n_all)
        t = call(n_all, nt, 1, cpus[:nt])
        reps = max(2, min(200, int(target_s / max(t, 1e-6))))
        legs[name] = (min(call(n_all, nt, reps, cpus[:nt]) for _ in range(3)), nt, reps)
    bestp, ntp, repsp = legs["physical"]
    r.update({"value": round(n_all * raw_per_chunk / bestp / 1e6, 1), "cores": ntp, "threads": ntp, "passes": repsp})
    if "logical" in legs:
        bl, ntl, _ = legs["logical"]
        r["value_all_logical_cpus"] = round(n_all * raw_per_chunk / bl / 1e6, 1)
        r["threads_all_logical_cpus"] = ntl
    r["scaling_vs_1thread"] = round(r["value"] / r["value_1thread"], 1)
    # the same threads on a sample small enough to stay in each core's L2 (~256 KB of samples a thread): what the cores do when
    # DRAM is out of the picture -- the gap between this and `value` is the host's memory system, not the harness
    n_small = min(n_all, ntp * max(1, (256 << 10) // raw_per_chunk))
    t = call(n_small, ntp, 1, firsts[:ntp])
    reps = max(2, min(20000, int(target_s / max(t, 1e-6))))
    bc = min(call(n_small, ntp, reps, firsts[:ntp]) for _ in range(3))
    r["value_cache_resident"] = round(n_small * raw_per_chunk / bc / 1e6, 1)
    r["scaling_cache_resident_vs_1thread"] = round(r["value_cache_resident"] / r["value_1thread"], 1)
    _, thr1 = cgroup_cpu_limit()
    r["cgroup_cpu_quota_cores"] = quota                    # None: no quota visible to this container
    if thr0 is not None and thr1 is not None:
        r["cgroup_throttled_periods_during_run"] = thr1 - thr0
    return r


def cpu_baseline(comp_np, offs_np, nchunks, codec_id, esz, chunk_len, target_s, what, n_one=8192, last_chunk_len=None):
    """The CPU path on a bounded sample of the same compressed chunks, on this host's cores.
    kind 'reference' = the real dblalock/sprintz AVX2/BMI2 code compiled into oracle/_ref (travels with
    the repo); 'port' = our scalar C restatement (oracle/liboracle.so) if that is absent.  The threads are pthreads
    inside liboracle.so (oracle/mt_bench.c) calling the decoder's chunk loop through a function pointer."""
    import numpy as np
    from tests.harness import ORACLE_SO
    lib, fn_name, _, kind = cpu_libs()
    if lib is None or not os.path.exists(ORACLE_SO):
        return None
    orc = C.CDLL(ORACLE_SO)
    mt = orc.oracle_mt_decompress_chunks
    mt.restype = C.c_double
    mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                   C.c_void_p, C.c_void_p]
    fn = C.cast(getattr(lib, fn_name), C.c_void_p)
    gap = 4096
    chunk_bytes = chunk_len * esz
    out = np.zeros(nchunks * chunk_bytes + ((os.cpu_count() or 1) + 1) * gap + 8192, np.uint8)
    elems = C.c_uint64(0)

    def call(n, nthreads, reps, cpus):
        arr = (C.c_int * nthreads)(*cpus) if cpus else None
        t = mt(fn, codec_id, esz, comp_np.ctypes.data, offs_np.ctypes.data, n, chunk_len, out.ctypes.data, gap, nthreads, reps, arr, C.byref(elems))
        want = n * chunk_len - ((chunk_len - last_chunk_len) if (last_chunk_len is not None and n == nchunks) else 0)
        assert t > 0 and elems.value == want, (t, elems.value, want)
        return t

    n_one = min(n_one, nchunks)
    r = time_cpu_mt(call, n_one, nchunks, chunk_bytes, target_s)
    r.update({"unit": "MB/s", "kind": kind,
              "sample": f"{nchunks} chunks ({nchunks * chunk_bytes / 1e6:.0f} MB raw) of this workload, {r['passes']} passes, best of 3, {r['cores']} pinned pthreads; 1-thread leg: {n_one} chunks",
              "sample_detail": f"{what} per chunk on this configuration's own compressed chunks, data in RAM; all-core legs: {nchunks} chunks "
                        f"({nchunks * chunk_bytes / 1e6:.0f} MB raw), one contiguous chunk range per pinned pthread (oracle/mt_bench.c), sustained "
                        f"over {r['passes']} passes, best of 3; `value`/`cores` = one thread per physical core, at most the container's CPU quota; 1-thread leg: first {n_one} chunks; "
                        f"value_cache_resident: the same threads on ~256 KB of samples each (L2-resident)"})
    return r


def cpu_baseline_huf0_chain(blocks_np, boffs_np, sizes_np, nchunks, esz, chunk_len, target_s, n_one=4096):
    """cfg4 on the host: Huff0 block -> Sprintz stream -> samples, chunk by chunk.  'reference' = the system
    libzstd's HUF_decompress (the coder the paper names) + the compiled reference decoder."""
    import numpy as np
    from tests.harness import ORACLE_SO
    if not os.path.exists(ORACLE_SO):
        return None
    orc = C.CDLL(ORACLE_SO)
    lib, _, dec_name, kind = cpu_libs()
    if lib is None:
        return None
    dec = C.cast(getattr(lib, dec_name), C.c_void_p)
    try:
        z = C.CDLL("libzstd.so.1")
        huf, hname = C.cast(z.HUF_decompress, C.c_void_p), "libzstd HUF_decompress"
    except (OSError, AttributeError):
        huf, hname, kind = C.cast(orc.oracle_huf0_decompress, C.c_void_p), "oracle_huf0_decompress", "port"
    mt = orc.oracle_mt_huf0_chain
    mt.restype = C.c_double
    mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                   C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ncpu = os.cpu_count() or 1
    gap = 4096
    chunk_bytes = chunk_len * esz
    out = np.zeros(nchunks * chunk_bytes + (ncpu + 1) * gap + 8192, np.uint8)
    scratch = np.zeros((ncpu + 1) * 65536, np.uint8)
    elems = C.c_uint64(0)

    def call(n, nthreads, reps, cpus):
        arr = (C.c_int * nthreads)(*cpus) if cpus else None
        t = mt(huf, dec, 1, esz, blocks_np.ctypes.data, boffs_np.ctypes.data, sizes_np.ctypes.data, n, chunk_len, scratch.ctypes.data,
               out.ctypes.data, gap, nthreads, reps, arr, C.byref(elems))
        assert t > 0 and elems.value == n * chunk_len, (t, elems.value)
        return t

    n_one = min(n_one, nchunks)
    r = time_cpu_mt(call, n_one, nchunks, chunk_bytes, target_s)
    r.update({"unit": "MB/s", "kind": kind,
              "sample": f"{nchunks} chunks ({nchunks * chunk_bytes / 1e6:.0f} MB raw) of this workload, {hname} + sprintz decode, {r['passes']} passes, best of 3, {r['cores']} pinned pthreads",
              "sample_detail": f"{hname} then sprintz_decompress_xff_16b per chunk on this configuration's own Huff0 blocks; all-core legs: {nchunks} chunks "
                        f"({nchunks * chunk_bytes / 1e6:.0f} MB raw), one chunk range per pinned pthread, {r['passes']} passes, best of 3; "
                        f"`value`/`cores` = one thread per physical core, at most the container's CPU quota; 1-thread leg: first {n_one} chunks"})
    return r


def streaming_probe():
    """tools/probes/mix_bw (if it was built): what a kernel that does nothing but coalesced 16-byte loads and non-temporal stores in
    the headline decoder's 6 : 17 read : write ratio moves on THIS box -- the practical ceiling next to the datasheet's 8 TB/s"""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "probes", "mix_bw")
    if not os.access(exe, os.X_OK):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    rates = [float(m.group(1)) for m in re.finditer(r"(?:decoder's mix|mix tiles)[^\n]*-> ([0-9.]+) TB/s", out)]
    copies = [float(m.group(1)) for m in re.finditer(r"tile copy[^\n]*-> ([0-9.]+) TB/s", out)]
    if not rates:
        return None
    r = {"GBps_best": round(max(rates) * 1e3, 1), "GBps_worst": round(min(rates) * 1e3, 1),
         "what": "tools/probes/mix_bw.hip: streaming kernels with the decoder's read : write mix -- grid-stride loops over 1024 / 4096 / 16384 "
                 "workgroups and one-tile-per-workgroup forms, nt and plain stores; copy_GBps_best = the same probe's 1 : 1 tile copy (the "
                 "calibration against MI355X_MICROARCH.md's 6.29 TB/s float4 copy)"}
    if copies:
        r["copy_GBps_best"] = round(max(copies) * 1e3, 1)
    return r


def roofline(algo_bytes, ms, kernel, extra=None):
    ach = algo_bytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
         "algorithmic_bytes_per_launch": int(algo_bytes), "kernel": kernel}
    if extra:
        r.update(extra)
    return r


def load_traffic(kernel_substr):
    """PMC-measured HBM bytes per launch of the named kernel, from the committed profile of THIS source tree's
    last profiled build (profiles/hbm_traffic.json; separate --pmc passes, gfx950 corrections): labelled, never silent"""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, None
    ent = t.get("entries", {}).get(kernel_substr) if "entries" in t else (t if kernel_substr in t.get("kernel", "") else None)
    if not ent:
        return None, None
    return ent.get("bytes_per_launch"), {"source": "profiles/" + ent.get("source", "hbm_traffic.json"), "build": ent.get("build", "unlabelled"),
                                         "nchunks": ent.get("nchunks"), "data": ent.get("data")}


def chain_traffic(name, algo_bytes, world):
    """PMC-measured HBM bytes of a cfg4 decode chain (profiles/hbm_traffic.json, entry "chain:<name>": the sum over the chain's
    kernels, last profiled build) next to the ratio the two-stage design implies: what the chain REALLY moves"""
    if world != 1:
        return {}
    t, label = load_traffic("chain:" + name)
    if not t:
        return {}
    return {"traffic": t, "traffic_ratio_measured": round(t / algo_bytes, 3), "traffic_measured_in_this_run": False, "traffic_source": label}


# ------------------------------------------------------------------------------------------ per-config legs
class Ctx:
    pass


def bench_rowmajor(cx, name, workload, codec, esz, ndims, chunk_len, nchunks_total, kind, step, huff0=False, strong=False):
    """one row-major configuration.  strong: nchunks_total is the WHOLE job, split over the ranks."""
    torch, dev, timer, args = cx.torch, cx.device, cx.timer, cx.args
    import sprintz_amd
    from sprintz_amd import _lib
    from synth import synth_torch
    lo, hi = (nchunks_total * cx.rank // cx.world, nchunks_total * (cx.rank + 1) // cx.world) if strong else (0, nchunks_total)
    n = hi - lo
    rows = chunk_len // ndims if chunk_len % ndims == 0 else None
    if rows is not None:
        x = synth_torch(kind, esz, n, rows, ndims, dev, seed=123, step=step, chunk0=lo + (0 if strong else cx.rank * nchunks_total))
    else:                                   # cfg3 at 1024 elements: chunks cut rows (the reference stores them raw anyway)
        from synth import synth_cut_rows
        x = synth_cut_rows(kind, esz, n, chunk_len, ndims, dev, seed=123 + cx.rank, step=step)
    if esz == 2:
        x = x.view(torch.int16)
    cd = sprintz_amd.ChunkedCodec(codec, esz, ndims, chunk_len, device=dev)
    src = cd._padded_view(x)
    ws = cd.workspace(n)
    dense = torch.empty(n * cd.slot_stride + _lib.READ_SLACK, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)

    def enc():                               # one call: for the fast encoder's shapes ONE launch builds the container too
        cd.compress_dense(src, x.numel(), ws, dense, offs)
    reps = args.config_reps if n * chunk_len * esz < (2 << 30) else max(3, args.config_reps // 4)
    ramp = min(args.ramp_ms, 100.0)          # per-configuration legs: at most 100 ms of ramp each (the headline: --ramp-ms)
    enc_ms = timer(enc, reps, ramp_ms=ramp)
    total = int(offs[-1].item())
    stream_bytes = int(ws["sizes"].to(torch.int64).sum().item())
    comp = dense[: total + _lib.READ_SLACK].clone()
    del dense
    out = torch.empty(n * chunk_len, dtype=x.dtype, device=dev)
    rets = torch.empty(n, dtype=torch.int64, device=dev)
    cd.decompress_into(comp, offs, n, out, rets)
    torch.cuda.synchronize()
    if not args.no_verify:
        assert torch.equal(out, x), f"{name}: GPU decode != input"
        assert bool((rets == chunk_len).all().item()), name
    dec_ms = timer(lambda: cd.decompress_into(comp, offs, n, out), reps, ramp_ms=ramp)
    raw = n * chunk_len * esz
    algo = stream_bytes + 8 * n + raw
    res = {"name": name, "workload": workload, "dtype": "u8" if esz == 1 else "u16", "ndims": ndims, "chunk_bytes": chunk_len * esz,
           "chunks": n, "chunks_all_ranks": nchunks_total if strong else n * cx.world, "scaling": "strong" if strong else "weak",
           "raw_bytes": raw, "ratio": round(raw / stream_bytes, 4),
           "decompress_ms": round(dec_ms, 4), "decompress_MBps": round(raw / dec_ms / 1e3, 1),
           "compress_ms": round(enc_ms, 4), "compress_MBps": round(raw / enc_ms / 1e3, 1),
           "roofline": roofline(algo, dec_ms, DECODE_KERNELS.get(name, DECODE_KERNELS.get(name.split("_")[0], "decode_fast_kernel<16, true, 8, 1, true, 0, false, 0>"))),
           "compress_roofline": roofline(raw + total + 12 * n, enc_ms, "sprintz_mi355x_compress_batch_dense: " + ENCODE_KERNELS.get(name, ENCODE_KERNELS.get(name.split("_")[0], "encode + scan + copy")),
                                         {"algorithmic": "raw samples in + dense container out + sizes/offsets (SURVEY 8d)",
                                          "traffic_ratio_by_design": round((raw + stream_bytes + 2 * total + 12 * n) / (raw + total + 12 * n), 3),
                                          "traffic_note": "the encoder writes slots, the compaction pass re-reads them and writes the dense container"})}
    if codec == "delta" and name in ("cfg1", "cfg3_10k") and not args.no_extras:
        # ---- round 6: the block-parallel delta kernels beside the lane-per-column / lane-per-chunk ones, same batch, same process
        # (SPRINTZ_OPT_BLK_KERNELS: 1 general-layout encoder, 2 general-layout decoder, 4 univariate encoder; the default mask is what won)
        ab = {"fields": "ms; *_lane: the lane-per-column / lane-per-chunk kernels of rounds 1 - 5; *_blk: the block-parallel kernels (csrc/encode_blk.h, decode_blk.h); "
                        "*_row: encode_blk + the column-group-sequential decoder (csrc/decode_row.h; general layout only); compress = the whole compress_batch_dense call"}
        dense2 = torch.empty(n * cd.slot_stride + _lib.READ_SLACK, dtype=torch.uint8, device=dev)
        offs2 = torch.empty(n + 1, dtype=torch.int64, device=dev)

        def enc2():
            cd.compress_dense(src, x.numel(), ws, dense2, offs2)
        try:
            for label, mask in (("lane", 0), ("blk", 7), ("row", 9)):
                _lib.check(_lib.set_option(_lib.OPT_BLK_KERNELS, mask))
                ab["compress_" + label] = round(timer(enc2, reps, ramp_ms=ramp), 4)
                cd.decompress_into(comp, offs, n, out, rets)
                torch.cuda.synchronize()
                assert torch.equal(out, x), f"{name}: decode != input on the {label} kernels"
                ab["decompress_" + label] = round(timer(lambda: cd.decompress_into(comp, offs, n, out), reps, ramp_ms=ramp), 4)
        finally:
            _lib.set_option(_lib.OPT_BLK_KERNELS, int(os.environ.get("SPRINTZ_MI355X_BLK_KERNELS", 9)))
        ab["default_mask"] = int(os.environ.get("SPRINTZ_MI355X_BLK_KERNELS", 9))
        res["block_parallel_ab"] = ab
        del dense2, offs2
    if n * chunk_len * esz < (64 << 20):
        res["note"] = ("launch-bound: %d chunks keep %d of the chip's 1024 SIMDs' worth of wavefronts busy; the time is one kernel's "
                       "end-to-end latency, not a bandwidth" % (n, min(1024, max(1, n * ndims // 64))))
    if huff0:
        # ---- Huff0 wire format (the paper's entropy coder; reader pinned against libzstd 1.4.8 blocks)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        z_buf = torch.zeros(int(_lib.huf0_bound(total, n)), dtype=torch.uint8, device=dev)
        z_offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
        z_tmp = torch.empty(int(_lib.huf0_tmp_bytes(n)), dtype=torch.uint8, device=dev)
        s_offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)          # byte-dense stream starts
        s_offs[1:] = torch.cumsum(ws["sizes"].to(torch.int64), 0)
        s_buf = torch.zeros(stream_bytes + _lib.READ_SLACK, dtype=torch.uint8, device=dev)
        z_rets = torch.empty(n, dtype=torch.int64, device=dev)

        def h_enc():
            _lib.check(_lib.huf0_compress_batch(comp.data_ptr(), offs.data_ptr(), ws["sizes"].data_ptr(), n, z_buf.data_ptr(),
                                                z_offs.data_ptr(), z_tmp.data_ptr(), st))

        zd_tmp = torch.empty(int(_lib.huf0_decode_tmp_bytes(n)), dtype=torch.uint8, device=dev)
        z_hint = [int(cd.slot_stride)]

        def h_dec():
            _lib.check(_lib.huf0_decompress_batch_hint(z_buf.data_ptr(), z_offs.data_ptr(), n, s_buf.data_ptr(), s_offs.data_ptr(),
                                                       z_rets.data_ptr(), zd_tmp.data_ptr(), z_hint[0], st))   # (hint: the writer's largest block)

        def chain():
            h_dec()
            cd.decompress_into(s_buf, s_offs, n, out)
        h_enc_ms = timer(h_enc, reps)
        z_hint[0] = int((z_offs[1:] - z_offs[:-1]).max().item())           # what the writer of the blocks knows: its largest block
        h_dec_ms = timer(h_dec, reps)
        chain_ms = timer(chain, reps)
        if not args.no_verify:
            assert torch.equal(z_rets, ws["sizes"].to(torch.int64)), "Huff0 decode: a block was rejected"
            assert torch.equal(out, x), "Huff0 -> Sprintz decode != input"
        hbytes = int(z_offs[-1].item())
        algo_chain = hbytes + 8 * n + 8 * n + raw                           # Huff0 blocks in + samples out + both offset tables
        moved_chain = algo_chain + 2 * stream_bytes                         # what the two-stage chain moves: the Sprintz streams out and in again
        res.update({"ratio": round(raw / hbytes, 4), "ratio_sprintz_only": round(raw / stream_bytes, 4),
                    "entropy_stage": "Huff0 wire format as of zstd 1.4.8 (HUF_compress-compatible blocks, one per chunk; reader and writer pinned against the system libzstd 1.4.8 -- the Huff0 revision inside the author's lzbench fork is not in the image)",
                    "decompress_ms": round(chain_ms, 4), "decompress_MBps": round(raw / chain_ms / 1e3, 1),
                    "huff0_decode_ms": round(h_dec_ms, 4), "sprintz_decode_ms": round(dec_ms, 4),
                    "compress_ms": round(enc_ms + h_enc_ms, 4), "compress_MBps": round(raw / (enc_ms + h_enc_ms) / 1e3, 1),
                    "huff0_encode_ms": round(h_enc_ms, 4),
                    "roofline": roofline(algo_chain, chain_ms, "Huff0 stage (tree passes + stream kernels, sprintz_mi355x_huf0_decompress_batch_ws) + sprintz decode; the Sprintz streams cross HBM between the two",
                                         {"algorithmic": "Huff0 blocks in + samples out + offset tables (SURVEY 8d); the intermediate Sprintz streams are NOT counted",
                                          "traffic_ratio_by_design": round(moved_chain / algo_chain, 3),
                                          **chain_traffic(name, algo_chain, cx.world),
                                          "huff0_decode_frac": round((hbytes + stream_bytes + 16 * n) / (h_dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})})
        if cx.rank == 0 and cx.world == 1 and not args.no_cpu_baseline:
            pt = huf0_private_trees(cx, comp, offs, ws["sizes"], n, timer, reps)
            if pt:
                res["huff0_private_trees"] = pt
        if cx.rank == 0 and not args.no_cpu_baseline:
            ns = min(n, max(64, CPU_SAMPLE_BYTES // (chunk_len * esz)))
            zo = z_offs[: ns + 1].cpu().numpy().astype("uint64")
            zb_h, sz_h = z_buf[: int(zo[ns]) + 64].cpu().numpy(), ws["sizes"][:ns].cpu().numpy().astype("uint32")
            cx.cpu_jobs.append((res, lambda: cpu_baseline_huf0_chain(zb_h, zo, sz_h, ns, esz, chunk_len, 1.0)))
        del z_buf, s_buf
    elif cx.rank == 0 and not args.no_cpu_baseline:
        ns = min(n, max(64, CPU_SAMPLE_BYTES // (chunk_len * esz)))
        o = offs[: ns + 1].cpu().numpy().astype("uint64")
        c_h = comp[: int(o[ns]) + 64].cpu().numpy()
        cx.cpu_jobs.append((res, lambda: cpu_baseline(c_h, o, ns, 1 if codec == "xff" else 0, esz, chunk_len, 1.0,
                                                      f"sprintz_decompress_{codec}_{8 * esz}b", n_one=max(64, (80 << 20) // (chunk_len * esz)))))
    res["_local"] = (raw, stream_bytes if not huff0 else hbytes, res["decompress_ms"], res["compress_ms"])
    del x, comp, out, src
    cd._ws = {}
    torch.cuda.empty_cache()
    return res


def bench_cfg5(cx, nrows_all=1 << 20, name="cfg5"):
    """BASELINE config 5: uint16 column-major, 32 variables, 1 M rows (64 MiB), FIRE, 160-row (10 KB) chunks =
    6554 chunks; contiguous row ranges per rank (strong scaling), sizes all-gathered.  cfg5_8m: the same shape with 8 M
    rows (512 MiB) -- the 1 M-row form is ONE 35 us launch, i.e. a latency; this one is the kernel's bandwidth."""
    torch, dev, timer, args = cx.torch, cx.device, cx.timer, cx.args
    import sprintz_amd
    from sprintz_amd import _lib
    from synth import synth_torch
    D, rpc, esz = 32, 160, 2
    nchunks_all = (nrows_all + rpc - 1) // rpc
    c_lo, c_hi = nchunks_all * cx.rank // cx.world, nchunks_all * (cx.rank + 1) // cx.world
    r_lo, r_hi = c_lo * rpc, min(c_hi * rpc, nrows_all)
    nrows, n = r_hi - r_lo, c_hi - c_lo
    # chunk c = rows [c*rpc, ...) of every column; generated chunk-wise (each chunk its own series), stored column-major
    full = synth_torch("walk", esz, n, rpc, D, dev, seed=123, step=8, chunk0=c_lo).view(n, rpc, D)
    cols = full.permute(2, 0, 1).reshape(D, n * rpc)[:, :nrows].contiguous()
    del full
    cd = sprintz_amd.ChunkedCodec("xff", esz, D, rpc * D, device=dev)
    batch = cd.compress_colmajor(cols)
    out = torch.empty((D, n * rpc), dtype=torch.uint16, device=dev)
    if not args.no_verify:
        assert torch.equal(cd.decompress_colmajor(batch, out=out).view(torch.int16), cols.view(torch.int16)), "cfg5: GPU decode != input"
    ws = cd.workspace(n)
    dense = torch.empty(n * cd.slot_stride + 16, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    st = cd._stream()

    def enc():                                # encode + container in one call (sprintz_mi355x_compress_batch_colmajor_dense)
        _lib.check(_lib.compress_batch_colmajor_dense(_lib.CODEC_XFF, esz, cols.data_ptr(), nrows, nrows, rpc, D, ws["slots"].data_ptr(),
                                                      cd.slot_stride, ws["sizes"].data_ptr(), ws["rets"].data_ptr(), dense.data_ptr(),
                                                      offs.data_ptr(), ws["tmp"].data_ptr(), st))

    def dec():
        _lib.check(_lib.decompress_batch_colmajor(_lib.CODEC_XFF, esz, batch.data.data_ptr(), batch.offsets.data_ptr(), n,
                                                  rpc, D, int(out.shape[1]), out.data_ptr(), None, st))
    reps = max(args.config_reps, 50)
    enc_ms, dec_ms = timer(enc, reps), timer(dec, reps)
    raw, sb = nrows * D * esz, batch.stream_bytes()
    res = {"name": name, "workload": f"uint16 colmajor, 32 variables, FIRE + bitpack + RLE, {nrows_all >> 20}M rows ({nrows_all >> 14} MiB), 160-row chunks",
           "dtype": "u16", "ndims": D, "chunk_bytes": rpc * D * esz, "chunks": n, "chunks_all_ranks": nchunks_all, "scaling": "strong",
           "raw_bytes": raw, "ratio": round(raw / sb, 4),
           "decompress_ms": round(dec_ms, 4), "decompress_MBps": round(raw / dec_ms / 1e3, 1),
           "compress_ms": round(enc_ms, 4), "compress_MBps": round(raw / enc_ms / 1e3, 1),
           "roofline": roofline(sb + 8 * n + raw, dec_ms, "decode_fast_kernel<16,FIRE,32,1,EXACT,0,CM=true>"),
           "compress_roofline": roofline(raw + sb + 12 * n, enc_ms, "sprintz_mi355x_compress_batch_colmajor_dense: encode_wide<..CM> (two columns per lane) + size scan + compaction copy (column-major sources keep the launches in a row: measured faster)",
                                         {"algorithmic": "raw samples in + dense container out + sizes/offsets (SURVEY 8d)"}),
           "note": ("64 MiB over %d chunks: one launch is %.0f us end to end, about half of it ramp-up and tail (launch-bound at this size; "
                    "cfg5_8m is the same shape at 512 MiB)" % (n, dec_ms * 1e3)) if nrows_all <= (1 << 20) else
                   "the column-major kernels at a size where a launch is bandwidth, not latency (BASELINE config 5 states 1M rows: entry cfg5)"}
    if cx.rank == 0 and not args.no_cpu_baseline:
        ns = min(n, max(64, CPU_SAMPLE_BYTES // (rpc * D * esz)))
        o = batch.offsets[: ns + 1].cpu().numpy().astype("uint64")
        c_h = batch.data[: int(o[ns]) + 64].cpu().numpy()
        last = (nrows - (n - 1) * rpc) * D if ns == n else None
        cx.cpu_jobs.append((res, lambda: cpu_baseline(c_h, o, ns, 1, esz, rpc * D, 1.0,
                                                      "sprintz_decompress_xff_16b (row-major flattening: the reference has no column-major entry)",
                                                      last_chunk_len=last)))
    res["_local"] = (raw, sb, res["decompress_ms"], res["compress_ms"])
    del cols, batch, out, dense
    cd._ws = {}
    torch.cuda.empty_cache()
    return res


def strong_scaling_prediction(cx, per):
    """{config: {"chunks": [n at 1, 2, 4, 8 ranks], "decode_ms": [...], "speedup": [...]}}: one rank's share of a fixed-size job measured on THIS
    GPU (rank 0's contiguous range; no collective is on the decode path) -- max-over-ranks time of an N-GPU run is this, if the GPUs are alike"""
    args = cx.args
    out = {}
    saved = (args.no_cpu_baseline, args.config_reps)
    args.no_cpu_baseline, args.config_reps = True, max(10, args.config_reps)
    try:
        for e in per:
            nm = e.get("name")
            if nm not in ("cfg4_10000", "cfg4_80000", "cfg5") or "error" in e:
                continue
            chunks, ms = [e["chunks"]], [e["decompress_ms"]]
            for N in (2, 4, 8):
                share = e["chunks"] // N
                if nm == "cfg5":
                    r = bench_cfg5(cx, nrows_all=share * 160, name=nm)
                else:
                    r = bench_rowmajor(cx, nm, "share", "xff", 2, 8, 5120, share, "walk", 8, huff0=True, strong=True)
                r.pop("_local", None)
                chunks.append(share)
                ms.append(r["decompress_ms"])
            out[nm] = {"chunks": chunks, "decode_ms": [round(v, 4) for v in ms], "speedup": [round(ms[0] / v, 2) for v in ms]}
    finally:
        args.no_cpu_baseline, args.config_reps = saved
    out["ranks"] = [1, 2, 4, 8]
    return out


def run_config(cx, name):
    if name == "cfg1":
        return bench_rowmajor(cx, "cfg1", "uint8 rowmajor, 1 variable, delta+zigzag+bitpack (low-dim layout), 1KB chunks, 512 MiB",
                              "delta", 1, 1, 1024, 524288, "walk", 2)
    if name == "cfg3_1k":
        return bench_rowmajor(cx, "cfg3_1k", "uint8 rowmajor, 80 variables, delta + bitpack + RLE, 1KB chunks (1024 el < one 16x80 group: "
                              "the reference stores such chunks verbatim, and so do we), 512 MiB", "delta", 1, 80, 1024, 524288, "walk", 2)
    if name == "cfg3_10k":
        return bench_rowmajor(cx, "cfg3_10k", "uint8 rowmajor, 80 variables, delta + bitpack + RLE, 10KB chunks (128 rows), 512 MiB",
                              "delta", 1, 80, 10240, 52429, "walk", 2)
    if name.startswith("cfg4_"):
        n = int(name.split("_")[1])
        return bench_rowmajor(cx, name, f"uint16 rowmajor, 8 variables, full Sprintz (FIRE + bitpack + RLE + Huff0), 10KB chunks, "
                              f"batch of {n} chunks sharded over the ranks", "xff", 2, 8, 5120, n, "walk", 8, huff0=True, strong=True)
    if name == "cfg5":
        return bench_cfg5(cx)
    if name == "cfg5_8m":
        return bench_cfg5(cx, nrows_all=8 << 20, name="cfg5_8m")
    raise ValueError(name)


def huf0_private_trees(cx, comp, offs, sizes, n, timer, reps):
    """cfg4's Huff0 stage on blocks as `HUF_compress` writes them -- a tree of its own in EVERY block (the writer of this
    library repeats one tree per 64-chunk segment, which its reader exploits): the first <= 4096 chunks' Sprintz streams
    coded by the host's libzstd, tiled to the batch, decoded on the GPU and compared with the streams.  None without libzstd."""
    import numpy as np
    torch, dev = cx.torch, cx.device
    try:
        z = C.CDLL("libzstd.so.1")
        z.HUF_compress.restype = C.c_size_t
        z.HUF_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.HUF_isError.restype = C.c_uint
        z.HUF_isError.argtypes = [C.c_size_t]
    except (OSError, AttributeError):
        return None
    from sprintz_amd import _lib
    nd = min(n, 4096)
    o = offs[: nd + 1].cpu().numpy().astype(np.int64)
    sz_h = sizes[:nd].cpu().numpy().astype(np.int64)
    host = comp[: int(o[nd])].cpu().numpy()
    blocks, tmp = [], np.zeros(1 << 17, np.uint8)
    for c in range(nd):
        st = np.ascontiguousarray(host[o[c]:o[c] + sz_h[c]])
        r = z.HUF_compress(tmp.ctypes.data, tmp.size, st.ctypes.data, st.size)
        blocks.append(st if (r == 0 or z.HUF_isError(r)) else tmp[:r].copy())
    k = max(1, n // nd)
    nt = nd * k
    bsz = np.array([b.size for b in blocks], np.int64)
    bo = np.zeros(nt + 1, np.int64)
    bo[1:] = np.cumsum(np.tile(bsz, k))
    oo = np.zeros(nt + 1, np.int64)
    oo[1:] = np.cumsum(np.tile(sz_h, k))
    one = torch.from_numpy(np.concatenate(blocks)).to(dev)
    d_blocks = torch.cat([one.repeat(k), torch.zeros(64, dtype=torch.uint8, device=dev)])
    plain_one = torch.from_numpy(np.concatenate([host[o[c]:o[c] + sz_h[c]] for c in range(nd)])).to(dev)
    d_bo, d_oo = torch.from_numpy(bo).to(dev), torch.from_numpy(oo).to(dev)
    d_out = torch.zeros(int(oo[-1]) + 64, dtype=torch.uint8, device=dev)
    d_rets = torch.empty(nt, dtype=torch.int64, device=dev)
    d_tmp = torch.empty(int(_lib.huf0_decode_tmp_bytes(nt)), dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def h_dec():
        _lib.check(_lib.huf0_decompress_batch_hint(d_blocks.data_ptr(), d_bo.data_ptr(), nt, d_out.data_ptr(), d_oo.data_ptr(),
                                                   d_rets.data_ptr(), d_tmp.data_ptr(), int(sz_h.max()), st))
    ms = timer(h_dec, reps)
    if not cx.args.no_verify:
        assert torch.equal(d_rets, torch.from_numpy(np.tile(sz_h, k)).to(dev)), "Huff0 decode (libzstd blocks): a block was rejected"
        assert torch.equal(d_out[: int(oo[-1])], plain_one.repeat(k)), "Huff0 decode (libzstd blocks) != the streams"
    hb, sb = int(bo[-1]), int(oo[-1])
    return {"blocks": "libzstd HUF_compress, one call per chunk (a tree of its own in every block); the first %d chunks' streams, "
                      "tiled %d times" % (nd, k),
            "chunks": nt, "huff0_bytes": hb, "stream_bytes": sb, "huff0_decode_ms": round(ms, 4),
            "huff0_decode_frac": round((hb + sb + 16 * nt) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def merge_over_ranks(cx, res):
    """strong-scaled legs: whole-job rate = all ranks' bytes / slowest rank's time"""
    from sprintz_amd.dist import max_over_ranks, sum_over_ranks
    raw, cbytes, dms, cms = res.pop("_local")
    if cx.world == 1:
        return res
    raw_all, cb_all = sum_over_ranks(raw, cx.device), sum_over_ranks(cbytes, cx.device)
    dms_all, cms_all = max_over_ranks(dms, cx.device), max_over_ranks(cms, cx.device)
    res.update({"job": {"raw_bytes_all_ranks": raw_all, "ratio": round(raw_all / cb_all, 4), "decompress_ms_max_rank": round(dms_all, 4),
                        "decompress_MBps": round(raw_all / dms_all / 1e3, 1), "compress_ms_max_rank": round(cms_all, 4),
                        "compress_MBps": round(raw_all / cms_all / 1e3, 1), "n_gpus": cx.world}})
    return res


# ------------------------------------------------------------------------------------------ main
def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: become `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py <same arguments>` -- one rank per GPU; rank 0 prints the JSON line."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["BENCH_SELF_LAUNCHED"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def dry_launch(args, world, rank):
    """the launch path without the workload: process group up, ranks all-gathered, one JSON line from rank 0"""
    import torch
    import torch.distributed as dist
    use_gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world and not os.environ.get("BENCH_ONE_DEVICE")
    backend = os.environ.get("BENCH_BACKEND", "nccl" if use_gpu else "gloo")
    seen = [rank]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
            t = torch.tensor([rank], device="cuda")
        else:
            dist.init_process_group(backend)
            t = torch.tensor([rank])
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        seen = [int(g.item()) for g in got]
    rccl = None
    json_fd = None
    if args.rccl:
        # (stdout carries the JSON line alone: RCCL prints its version banner there when the first communicator comes up)
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        # the product's own exchange (csrc/comm.cpp: ncclAllGather behind the C-ABI) on a real communicator of `world` ranks:
        # rank r contributes 1000 + r bytes, every rank must see all of them and derive the same bases
        one_device = bool(os.environ.get("BENCH_ONE_DEVICE")) and torch.cuda.is_available()
        assert one_device or (use_gpu and (world == 1 or backend == "nccl")), "--rccl needs one visible GPU per rank (or BENCH_ONE_DEVICE=1: the fall-back drill)"
        from sprintz_amd.dist import LayoutGather
        dev = torch.device("cuda", 0 if one_device else int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        # BENCH_ONE_DEVICE=1: the FALL-BACK DRILL -- N ranks on one GPU, gloo bootstrap, the library's communicator attempted all the same:
        # RCCL refuses a second rank on the device, every rank must say so loudly and take torch.distributed together, the layout must still be right
        g = LayoutGather(dev, prefer_c_abi="force" if one_device else True)
        if world > 1 and g.comm is None:
            print(f"[bench rank {rank}] LAYOUT GATHER FELL BACK from rccl-c-abi (sprintz_mi355x_gather_layout) to {g.backend}; "
                  f"the C-ABI communicator did not come up: {g.c_abi_error}", file=sys.stderr, flush=True)
            if args.strict_rccl:
                raise SystemExit(f"--strict-rccl: rank {rank}: layout gather is not rccl-c-abi ({g.c_abi_error})")
        if g.comm is None and world > 1 and one_device:
            g.device, g.all = torch.device("cpu"), torch.zeros(world, dtype=torch.int64)     # gloo carries CPU tensors
        lay = g.layout(1000 + rank)
        want = [1000 + r for r in range(world)]
        assert lay.rank_bytes == want, (lay.rank_bytes, want)
        assert lay.rank_base == [sum(want[:r]) for r in range(world)] and lay.total_bytes == sum(want)
        rccl = {"backend": g.backend, "ranks_seen": g.ranks_seen, "c_abi_error": g.c_abi_error, "rank_bytes": lay.rank_bytes}
        assert world == 1 or g.comm is not None or one_device, f"comm.cpp could not bring RCCL up: {g.c_abi_error}"
        g.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        line = json.dumps({"dry_launch": True, "n_gpus": args.gpus, "world": world, "ranks": seen, "backend": backend if world > 1 else None,
                           "self_launched": bool(os.environ.get("BENCH_SELF_LAUNCHED")), "rccl_c_abi": rccl})
        if json_fd is not None:
            sys.stdout.flush()
            try:
                C.CDLL(None).fflush(None)
            except Exception:
                pass
            os.write(json_fd, (line + "\n").encode())
        else:
            print(line, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                        # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    assert world == args.gpus, (f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}: start it with --nproc-per-node {args.gpus}, "
                                f"or with no launcher at all (bench.py then launches its own ranks)")
    if args.dry_launch:
        return dry_launch(args, world, rank)
    import torch
    import torch.distributed as dist

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # stdout carries exactly ONE line, the JSON: whatever libraries print while the job runs (RCCL's version banner on
    # the first communicator, for one) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        os.write(json_fd, (json.dumps(obj) + "\n").encode())
    # (test aid: BENCH_ONE_DEVICE=1 BENCH_BACKEND=gloo runs N ranks on ONE GPU to exercise the sharding / strong-scaling /
    #  merge logic on a single-GPU box; RCCL itself refuses two ranks on one device)
    if os.environ.get("BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import sprintz_amd
    from sprintz_amd import _lib
    from sprintz_amd.dist import LayoutGather, max_over_ranks, sum_over_ranks
    from synth import synth_torch

    cx = Ctx()
    cx.torch, cx.device, cx.world, cx.rank, cx.args, cx.timer = torch, device, world, rank, args, Timer(torch, min(args.ramp_ms, 50.0))      # every timed leg starts on a warm clock (50 ms of its own launches; the rowmajor legs 100)
    cx.cpu_jobs = []                      # rank 0's host legs: run after the last collective (see the end of main)
    numa, cx.affinity_before = pin_to_gpu_numa_node(torch, local_rank) if world > 1 else ({"numa": "single rank: not pinned"}, None)

    if args.only:
        res = merge_over_ranks(cx, run_config(cx, args.only))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            restore_affinity(cx)
            run_cpu_jobs(cx)
            emit(res)
        return

    codec_name, esz, ndims, chunk_len = "xff", 2, 8, 5120
    rows = chunk_len // ndims
    nchunks = args.nchunks
    chunk_bytes = chunk_len * esz
    kind, step = DATA_KINDS[args.data]

    # ---------------- setup (untimed): data, GPU compress, global layout
    x = synth_torch(kind, esz, nchunks, rows, ndims, device, seed=123, step=step, chunk0=rank * nchunks).view(torch.int16)
    codec = sprintz_amd.ChunkedCodec(codec_name, esz, ndims, chunk_len, device=device)
    src_padded = codec._padded_view(x)
    ws = codec.workspace(nchunks)
    dense = torch.empty(nchunks * codec.slot_stride + _lib.READ_SLACK, dtype=torch.uint8, device=device)
    offsets = torch.empty(nchunks + 1, dtype=torch.int64, device=device)
    gather = LayoutGather(device)            # RCCL behind the C-ABI when it comes up, torch.distributed otherwise
    timer = cx.timer
    # ---------------- N > 1: say EARLY and LOUDLY what carries the one exchange of the write path
    gather_fallback = None
    if world > 1 and gather.comm is None:
        gather_fallback = f"{gather.backend}; the C-ABI communicator did not come up: {gather.c_abi_error}"
        print(f"[bench rank {rank}] LAYOUT GATHER FELL BACK from rccl-c-abi (sprintz_mi355x_gather_layout) to {gather_fallback}", file=sys.stderr, flush=True)
        if args.strict_rccl:
            raise SystemExit(f"--strict-rccl: rank {rank}: layout gather is not rccl-c-abi ({gather_fallback})")

    def compress_step():                     # encode + container (one launch) -> (N > 1) all-gather of byte counts: the whole write path
        codec.compress_dense(src_padded, x.numel(), ws, dense, offsets)
        gather.gather_async(offsets[nchunks:])

    def compress_two_launches():             # round 2's path: encode into slots, then size scan + copy
        codec.compress_to_slots(src_padded, x.numel(), ws)
        codec.compact(ws, nchunks, dense, offsets)
    compress_2l_ms = timer(compress_two_launches, 5)
    compress_ms = timer(compress_step, 5)
    compress_ms = max_over_ranks(compress_ms, device)
    total_comp = int(offsets[-1].item())
    stream_bytes = int(ws["sizes"].to(torch.int64).sum().item())
    comp = dense[: total_comp + _lib.READ_SLACK].clone()
    del dense
    layout = gather.layout(total_comp)       # bases of every rank's container in the job-wide one
    assert gather.ranks_seen == world and len(layout.rank_bytes) == world and all(b > 0 for b in layout.rank_bytes), \
        f"rank {rank}: the layout exchange saw {gather.ranks_seen} of {world} ranks: {layout.rank_bytes}"
    out = torch.empty(nchunks * chunk_len, dtype=torch.int16, device=device)
    rets = torch.empty(nchunks, dtype=torch.int64, device=device)

    # correctness of what is about to be timed
    codec.decompress_into(comp, offsets, nchunks, out, rets)
    torch.cuda.synchronize()
    if not args.no_verify:
        assert torch.equal(out, x), "GPU decode != input"
        assert bool((rets == chunk_len).all().item())

    # ---------------- untimed: let DVFS settle on this workload (steady-state serving is what is measured)
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(20):
            codec.decompress_into(comp, offsets, nchunks, out)
        torch.cuda.synchronize()

    # ---------------- timed region
    for _ in range(args.warmup):
        codec.decompress_into(comp, offsets, nchunks, out)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # ONE pair of HIP events around the K launches (torch's current stream == the launch stream): their elapsed time / K is the average
    # launch duration, back to back as a serving loop issues them.  (Through round 5 every launch had its own pair: the 2 K event records
    # between the kernels cost the wall clock ~3 % -- 0.4177 against 0.4035 ms -- which the line's roofline.frac now follows.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        codec.decompress_into(comp, offsets, nchunks, out)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wall_ranks = [wall]
    if world > 1:                                          # every rank's own clock beside the maximum the line is quoted on
        tw = torch.tensor([wall], dtype=torch.float64, device=device)
        got = [torch.zeros_like(tw) for _ in range(world)]
        dist.all_gather(got, tw)
        wall_ranks = [float(g.item()) for g in got]
    wall = max_over_ranks(wall, device)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps                       # HIP-event average launch duration

    total_raw = sum_over_ranks(nchunks * chunk_bytes, device)
    total_stream = sum_over_ranks(stream_bytes, device)
    value = total_raw * args.steps / wall / 1e6
    algo_bytes = stream_bytes + 8 * nchunks + nchunks * chunk_bytes
    traffic, traffic_label = load_traffic("decode_fast_kernel<16, true, 8")
    if traffic_label and not (traffic_label.get("nchunks") == nchunks and traffic_label.get("data") == args.data):
        traffic, traffic_label = None, None
    result = {
        "metric": METRIC,
        "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u16",
        "data": f"synthetic ({args.data}: SURVEY 8d generator, splitmix64 seeded per chunk, seed 123; tools/synth.py == oracle/synth.c)",
        "config": {"workload": "cfg2: uint16 rowmajor, 8 variables, FIRE predictor + bitpack + RLE "
                               "(sprintz_xff_16b), 10KB chunks",
                   "chunks_per_gpu": nchunks, "chunk_bytes": chunk_bytes, "raw_bytes_per_gpu": nchunks * chunk_bytes,
                   "sharding": f"chunks x{world}, no data-path collective"},
        "ratio": round(total_raw / total_stream, 4),
        "compress_MBps": round(total_raw / (compress_ms * 1e-3) / 1e6, 1),
        "compress": {"ms_per_step_max_rank": round(compress_ms, 4),
                     "what": "sprintz_mi355x_compress_batch_dense: ONE launch, the encoder's workgroups find their place in the container by a chained "
                             "scan and copy their own chunks into it (csrc/compact_tail.h)" + (" + all-gather of per-rank byte counts" if world > 1 else ""),
                     "two_launch_ms_this_rank": round(compress_2l_ms, 4), "layout_gather": gather.backend,
                     "roofline_frac": round((nchunks * chunk_bytes + total_comp + 12 * nchunks) / (compress_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "roofline_algorithmic": "raw samples in + dense container out + sizes/offsets, this rank"},
        "ms_per_step_ranks": [round(w / args.steps * 1e3, 4) for w in wall_ranks],
        "kernel_ms": round(kernel_ms, 4),
        # `achieved` / `frac`: this rank's algorithmic bytes per launch over the line's own ms_per_step (the wall clock the driver can
        # re-derive: bytes / ms_per_step / 8 TB/s); the HIP-event average of the same launches beside it as *_kernel_events
        "roofline": {"bound": "hbm", "achieved": round(algo_bytes / (wall / args.steps) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(algo_bytes / (wall / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     "achieved_kernel_events": round(algo_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                     "frac_kernel_events": round(algo_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_measured_in_this_run": False,      # (PMC passes are separate rocprofv3 runs: the figure is the committed profile's, labelled below; null if none matches)
                     "traffic_source": traffic_label, "algorithmic_bytes_per_launch": algo_bytes,
                     "kernel": "decode_fast_kernel<16,FIRE,8,1,EXACT>"},
        "container_bytes_all_ranks": layout.total_bytes, "rank_bases": layout.bases[:8], "rccl_ranks_seen": gather.ranks_seen,
        "layout_gather_fallback": gather_fallback,
        # what the multi-GPU sharding of every configuration has run on so far (a reader of this line alone should know)
        "configs_8gpu_sharding": "gloo on CPU + 8 ranks on one device only; never on 8 devices" if world == 1 else f"this run: {world} ranks, {gather.backend}",
        "host": {**host_description(), **numa},
    }

    if rank == 0 and not args.no_extras:
        sp = streaming_probe()
        if sp:
            moved = (traffic or algo_bytes) / (kernel_ms * 1e-3) / 1e9
            sp["decoder_GBps_of_traffic"] = round(moved, 1)
            sp["decoder_vs_best_streaming"] = round(moved / sp["GBps_best"], 3)
            result["roofline"]["streaming_probe"] = sp
    if not args.no_extras:
        result.update(headline_extras(cx, codec, x, comp, offsets, ws, out, nchunks, total_comp, chunk_len, ndims, esz, wall / args.steps * 1e3))

    if rank == 0 and not args.no_cpu_baseline:
        ns = min(nchunks, max(64, CPU_SAMPLE_BYTES // chunk_bytes))
        offs_np = offsets[: ns + 1].cpu().numpy().astype("uint64")
        comp_np = comp[: int(offs_np[ns]) + 64].cpu().numpy()
        cx.cpu_jobs.append((result, lambda: cpu_baseline(comp_np, offs_np, ns, 1, esz, chunk_len, args.cpu_seconds, "sprintz_decompress_xff_16b")))
    del x, comp, out, src_padded
    codec._ws = {}
    torch.cuda.empty_cache()

    # ---------------- the headline shape on SURVEY 8d's other generators (G0 uniform, G1 +-300, G2 walk + flat spans)
    if not args.no_sweep:
        result["data_sweep"] = data_sweep(cx, codec, nchunks, rows, ndims, chunk_len, esz)

    # ---------------- every other BASELINE configuration, same process, same clock
    names = [] if args.configs == "none" else (ALL_CONFIGS if args.configs == "all" else [c for c in args.configs.split(",") if c])
    per = [{"name": "cfg2", "workload": result["config"]["workload"], "scaling": "weak", "see": "top-level fields of this line"}]
    for nm in names:
        try:
            per.append(merge_over_ranks(cx, run_config(cx, nm)))
        except Exception as e:      # a failing leg must not cost the headline line
            if world > 1:
                raise
            per.append({"name": nm, "error": f"{type(e).__name__}: {e}"})
    result["per_config"] = per
    # ---------------- what ONE GPU takes for one rank's share of the strong-scaled configurations at 2 / 4 / 8 ranks: the prediction a
    # measured N-GPU run is to be read against (cfg4 at BASELINE's 10 000 chunks is three serial latencies: 8 GPUs cannot buy 8x)
    if world == 1 and names and not args.no_extras:
        result["strong_scaling_prediction"] = strong_scaling_prediction(cx, per)

    # ---------------- the last collective is behind us: the ranks part, and ONLY THEN does rank 0 spend its minute of CPU legs
    # (no rank waits inside RCCL while another one times the host)
    gather.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    restore_affinity(cx)
    run_cpu_jobs(cx)
    finish_and_emit(result, emit)


def run_cpu_jobs(cx):
    for target, job in cx.cpu_jobs:
        r = job()
        if r is not None:
            target["cpu_baseline"] = r
    cx.cpu_jobs = []


SUMMARY_FIELDS = "dec_ms, dec_frac, enc_ms, enc_frac, cpu_allcore_MBps, cpu_1thread_MBps"
LINE_LIMIT = 10000          # the driver's parser took a 13.4 KB line and not a 24.5 KB one: stay well inside
LONG_TEXT_KEYS = ("what", "note", "sample_detail", "algorithmic", "traffic_note", "blocks", "reference_build", "parity", "kernel_detail",
                  "threads_limit", "entropy_stage")


def compact(obj, depth=0):
    """the line the driver parses: every number, none of the prose (that lives in bench_full.json)"""
    if isinstance(obj, dict):
        return {k: compact(v, depth + 1) for k, v in obj.items() if k not in LONG_TEXT_KEYS and k != "per_config"}
    if isinstance(obj, list):
        return [compact(v, depth + 1) for v in obj]
    return obj


def finish_and_emit(result, emit):
    """per_config_summary (LAST key: a reader who keeps only the tail of the line sees every configuration), the full record to
    bench_full.json + stderr, the compact line to stdout"""
    per = result["per_config"]
    summ = {"cfg2": [result["ms_per_step"], result["roofline"]["frac"], result["compress"]["ms_per_step_max_rank"], result["compress"]["roofline_frac"],
                     result.get("cpu_baseline", {}).get("value"), result.get("cpu_baseline", {}).get("value_1thread")]}
    for e in per[1:]:
        if "error" in e:
            summ[e["name"]] = "error"
            continue
        cb = e.get("cpu_baseline") or {}
        j = e.get("job") or {}
        summ[e["name"]] = [j.get("decompress_ms_max_rank", e["decompress_ms"]), e["roofline"]["frac"], j.get("compress_ms_max_rank", e["compress_ms"]),
                           e["compress_roofline"]["frac"], cb.get("value"), cb.get("value_1thread")]
    result["per_config_summary"] = {"fields": SUMMARY_FIELDS, **summ}
    # the driver's record keeps `config`, `roofline` and `cpu_baseline` whole and drops every other extra key: the per-configuration
    # fractions ride inside `roofline` so that a reader of BENCH_rNN.json's `parsed` alone sees them
    result["roofline"]["per_config_fracs"] = {"fields": "dec_ms, dec_frac, enc_ms, enc_frac (algorithmic bytes / time / 8 TB/s; cfg4: the whole chain)",
                                        **{k: (v[:4] if isinstance(v, list) else v) for k, v in summ.items()}}
    result["config"]["multi_gpu_evidence"] = result.get("configs_8gpu_sharding")
    if result.get("layout_gather_fallback"):
        result["config"]["layout_gather_fallback"] = result["layout_gather_fallback"]
    full_path = os.path.join(ROOT, "bench_full.json")
    result["full_record"] = "bench_full.json next to bench.py (per_config entries, the prose of every field); also on stderr"
    result["per_config_summary"] = result.pop("per_config_summary")          # stays the last key
    try:
        with open(full_path, "w") as f:
            json.dump(result, f)
            f.write("\n")
    except OSError as e:
        print(f"bench_full.json not written: {e}", file=sys.stderr)
    print("BENCH_FULL " + json.dumps(result), file=sys.stderr, flush=True)
    line = compact(result)
    # the driver's parser needs the line short: should it ever outgrow the limit, the extras go first (they stay in bench_full.json) --
    # never the contract fields, never an exception instead of a line
    for k in ("real_data", "single_call_threads", "any_ndims", "transforms", "online_coders", "pcie_inclusive", "query", "huffman", "strong_scaling_prediction", "data_sweep"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        if k in line:
            line.pop(k)
            line.setdefault("dropped_from_line", []).append(k)
    emit(line)


def native_threads_leg(counts=(1, 8, 64)):
    """tools/probes/single_call_mt.cpp built with g++ into a temp dir and run as a subprocess -> {threads: calls_per_s}."""
    import shutil
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "tools", "probes", "single_call_mt.cpp")
    lib = os.path.join(ROOT, "sprintz_amd", "libsprintz_mi355x.so")
    if not shutil.which("g++") or not os.path.exists(src):
        return {"skipped": "no g++ or no tools/probes/single_call_mt.cpp"}
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "single_call_mt")
        try:
            subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, src, "-ldl", "-lpthread"], check=True, timeout=120,
                           stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            r = subprocess.run([exe, lib] + [str(c) for c in counts], check=True, timeout=120, stdin=subprocess.DEVNULL,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        except (subprocess.CalledProcessError, subprocess.TimeoutExpired) as e:
            return {"failed": str(e)[:200]}
    out = {}
    for m in re.finditer(r"(\d+) threads:\s+(\d+) calls/s", r.stdout):
        out[m.group(1)] = {"calls_per_s": int(m.group(2))}
    if "FAILED" in r.stdout or not out:
        return {"failed": r.stdout[-200:]}
    return out


def data_sweep(cx, codec, nchunks, rows, ndims, chunk_len, esz):
    """SURVEY 8d's generators on the headline shape, same launches as the headline: G0 uniform (the paper's worst case,
    results.tex:142-146 -- every field 16 bits, stream > input), G1 walk +-300, G2 walk + flat spans (every 4th 64-row span
    constant: the encoder's RLE state machine and the decoder's run replay, sprintz_xff_rle.cpp:828-958, are on the path).
    Each: one-launch compress, decode checked against the input, then both timed.  -> {kind: [dec_ms, dec_frac, enc_ms, enc_frac, ratio]}"""
    torch, dev, timer, args = cx.torch, cx.device, cx.timer, cx.args
    from sprintz_amd import _lib
    from sprintz_amd.dist import max_over_ranks
    from synth import synth_torch
    out = {"fields": "dec_ms, dec_frac, enc_ms, enc_frac, ratio", "chunks_per_gpu": nchunks}
    raw = nchunks * chunk_len * esz
    for name in ("uniform", "walk300", "walkflat"):
        kind, step = DATA_KINDS[name]
        x = synth_torch(kind, esz, nchunks, rows, ndims, dev, seed=123, step=step, chunk0=cx.rank * nchunks).view(torch.int16)
        src = codec._padded_view(x)
        ws = codec.workspace(nchunks)
        dense = torch.empty(nchunks * codec.slot_stride + _lib.READ_SLACK, dtype=torch.uint8, device=dev)
        offs = torch.empty(nchunks + 1, dtype=torch.int64, device=dev)
        o = torch.empty(nchunks * chunk_len, dtype=torch.int16, device=dev)
        rets = torch.empty(nchunks, dtype=torch.int64, device=dev)
        enc_ms = timer(lambda: codec.compress_dense(src, x.numel(), ws, dense, offs), 10)
        total = int(offs[-1].item())
        sb = int(ws["sizes"].to(torch.int64).sum().item())
        codec.decompress_into(dense, offs, nchunks, o, rets)
        torch.cuda.synchronize()
        if not args.no_verify:
            assert torch.equal(o, x), f"data_sweep {name}: GPU decode != input"
            assert bool((rets == chunk_len).all().item()), name
        dec_ms = timer(lambda: codec.decompress_into(dense, offs, nchunks, o), 20)
        enc_ms, dec_ms = max_over_ranks(enc_ms, dev), max_over_ranks(dec_ms, dev)
        out[name] = [round(dec_ms, 4), round((sb + 8 * nchunks + raw) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     round(enc_ms, 4), round((raw + total + 12 * nchunks) / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), round(raw / sb, 4)]
        del x, src, dense, o
        codec._ws = {}
        torch.cuda.empty_cache()
    return out


def pin_to_gpu_numa_node(torch, local_rank):
    """one rank per GPU: run this rank's host threads on the CPUs of the GPU's NUMA node (sysfs: the PCI device's
    local_cpulist), so that launches and pinned staging stay node-local.  -> (description for `host`, affinity to restore)"""
    try:
        before = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"numa": "no sched_getaffinity"}, None
    info = {"numa_node": None, "cpus_pinned": None}
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        base = f"/sys/bus/pci/devices/{bdf}"
        info["pci"] = bdf
        info["numa_node"] = int(open(base + "/numa_node").read())
        cpus = set()
        for part in open(base + "/local_cpulist").read().strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= set(before)
        if cpus and len(cpus) < len(before):
            os.sched_setaffinity(0, cpus)
            info["cpus_pinned"] = len(cpus)
    except Exception as e:                       # no sysfs entry, no such attribute: run unpinned and say so
        info["numa_error"] = f"{type(e).__name__}: {e}"[:120]
    return info, before


def restore_affinity(cx):
    if getattr(cx, "affinity_before", None):
        try:
            os.sched_setaffinity(0, cx.affinity_before)
        except OSError:
            pass


def headline_extras(cx, codec, x, comp, offsets, ws, out, nchunks, total_comp, chunk_len, ndims, esz, step_ms):
    """secondary numbers on the headline batch: both Huffman containers, query on compressed data, the
    single-call latency lzbench would see, and the PCIe-inclusive rate of the host-buffer entry points"""
    import numpy as np
    torch, device, args, timed = cx.torch, cx.device, cx.args, cx.timer
    from sprintz_amd import _lib
    from sprintz_amd.codec import CompressedBatch, huf_compress
    from sprintz_amd.dist import sum_over_ranks
    chunk_bytes = chunk_len * esz
    total_raw = sum_over_ranks(nchunks * chunk_bytes, device)
    res = {}
    cb = CompressedBatch(comp, offsets, ws["sizes"], nchunks, x.numel(), chunk_len, ndims)
    hb = huf_compress(cb)
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    h_buf = torch.zeros(int(_lib.huf_bound(total_comp, nchunks)), dtype=torch.uint8, device=device)
    h_offs = torch.empty(nchunks + 1, dtype=torch.int64, device=device)
    h_tabs = torch.empty(((nchunks + 63) // 64) * 128, dtype=torch.uint8, device=device)
    h_tmp = torch.empty(int(_lib.huf_tmp_bytes(nchunks)), dtype=torch.uint8, device=device)
    d_buf = torch.zeros(total_comp + _lib.READ_SLACK + 16 * nchunks, dtype=torch.uint8, device=device)
    d_offs = torch.empty(nchunks + 1, dtype=torch.int64, device=device)
    d_sizes = torch.empty(nchunks, dtype=torch.int32, device=device)

    def huf_enc():
        _lib.check(_lib.huf_compress_batch(comp.data_ptr(), offsets.data_ptr(), ws["sizes"].data_ptr(), nchunks, h_buf.data_ptr(),
                                           h_offs.data_ptr(), h_tabs.data_ptr(), h_tmp.data_ptr(), st))

    def huf_dec():
        _lib.check(_lib.huf_decompress_batch(h_buf.data_ptr(), h_offs.data_ptr(), h_tabs.data_ptr(), nchunks, 16, d_buf.data_ptr(),
                                             d_buf.numel() - _lib.READ_SLACK, d_offs.data_ptr(), d_sizes.data_ptr(), None,
                                             h_tmp.data_ptr(), st))
    huf_enc_ms = timed(huf_enc, 5, 1)
    huf_dec_ms = timed(huf_dec, 5, 1)
    if not args.no_verify:
        assert int(h_offs[-1].item()) + h_tabs.numel() == hb.total_bytes(), "Huffman container size differs between runs"
        codec.decompress_into(d_buf, d_offs, nchunks, out)
        torch.cuda.synchronize()
        assert torch.equal(out, x), "Huffman -> Sprintz decode != input"
    huf_bytes = sum_over_ranks(hb.total_bytes(), device)
    del h_buf, d_buf, hb
    res["huffman_stage_own_container"] = {
        "ratio": round(total_raw / huf_bytes, 4), "encode_ms": round(huf_enc_ms, 3), "decode_ms": round(huf_dec_ms, 3),
        "chain_decompress_MBps": round(nchunks * chunk_bytes / ((huf_dec_ms + step_ms) * 1e-3) / 1e6, 1),
        "parity": "OWN FORMAT, UNPINNED (no Huffman coder in the reference tree); the pinned stage is per_config cfg4_* (Huff0 wire format)"}

    # ---------------- query on compressed data: per-column sum fused into the decode
    q_part = torch.empty((nchunks, ndims), dtype=torch.int64, device=device)
    q_res = torch.empty(ndims, dtype=torch.int64, device=device)

    def query_reduce_only():
        _lib.check(_lib.query_batch(_lib.CODEC_XFF, esz, comp.data_ptr(), offsets.data_ptr(), nchunks, chunk_len, ndims,
                                    _lib.QUERY_SUM, 0, 0, None, q_part.data_ptr(), None, st))
        _lib.check(_lib.query_reduce(_lib.QUERY_SUM, q_part.data_ptr(), nchunks, ndims, q_res.data_ptr(), st))

    def query_materialize():
        _lib.check(_lib.query_batch(_lib.CODEC_XFF, esz, comp.data_ptr(), offsets.data_ptr(), nchunks, chunk_len, ndims,
                                    _lib.QUERY_SUM, 1, 0, out.data_ptr(), q_part.data_ptr(), None, st))
        _lib.check(_lib.query_reduce(_lib.QUERY_SUM, q_part.data_ptr(), nchunks, ndims, q_res.data_ptr(), st))
    query_ms = timed(query_reduce_only, 5, 1)
    if not args.no_verify:
        want = x.view(torch.int16).to(torch.int64).bitwise_and(0xffff).view(-1, ndims).sum(dim=0)
        assert torch.equal(q_res, want), "query(sum) != column sums of the input"
    query_mat_ms = timed(query_materialize, 5, 1)
    res["query_on_compressed"] = {"op": "sum", "reduce_only_ms": round(query_ms, 3),
                                  "reduce_only_MBps": round(nchunks * chunk_bytes / (query_ms * 1e-3) / 1e6, 1),
                                  "materialize_ms": round(query_mat_ms, 3)}

    if cx.rank == 0:
        # ---------------- the drop-in single-call symbols as lzbench drives them: one 10 KB chunk per call, host pointers
        o = offsets[:2].cpu().numpy()
        one = np.ascontiguousarray(comp[int(o[0]): int(o[1]) + 64].cpu().numpy())
        raw1 = np.ascontiguousarray(x[:chunk_len].view(torch.int16).cpu().numpy().view(np.uint16))
        dst = np.zeros(chunk_len + 64, np.uint16)
        cdst = np.zeros(chunk_len * 3 // 2 + 64, np.int16)
        dfn, cfn = _lib.decompress[("xff", 2)], _lib.compress[("xff", 2)]
        for _ in range(20):
            dfn(one.ctypes.data, dst.ctypes.data)
            cfn(raw1.ctypes.data, chunk_len, cdst.ctypes.data, ndims, 1)
        lat_d, lat_c = [], []
        for _ in range(300):
            t0 = time.perf_counter(); r = dfn(one.ctypes.data, dst.ctypes.data); lat_d.append(time.perf_counter() - t0)
            assert r == chunk_len
            t0 = time.perf_counter(); cfn(raw1.ctypes.data, chunk_len, cdst.ctypes.data, ndims, 1); lat_c.append(time.perf_counter() - t0)
        assert np.array_equal(dst[:chunk_len], raw1)
        lat_d.sort(); lat_c.sort()
        res["single_call_latency_10KB"] = {
            "decompress_us_median": round(lat_d[150] * 1e6, 1), "decompress_us_p10": round(lat_d[30] * 1e6, 1),
            "compress_us_median": round(lat_c[150] * 1e6, 1), "compress_us_p10": round(lat_c[30] * 1e6, 1),
            "what": "sprintz_decompress_xff_16b / sprintz_compress_xff_16b on host buffers through ctypes: memcpy into the thread's MAPPED staging "
                    "buffer, ONE launch of the one-workgroup-per-chunk kernel (decode_lat.h / encode_lat.h) reading and writing that buffer directly "
                    "and ending with the call's ticket in a mapped host word the caller polls, memcpy out; no staging kernel, no runtime wait, no copy "
                    "engine, no memset, no hipMalloc per call"}
        # ---------------- BASELINE configs[0]'s shape through the same symbols: uint8, 1 variable, 1 KB chunks (the low-dim layout)
        from synth import synth_numpy
        u1 = synth_numpy("walk", 1, 1, 1024, 1, seed=123, step=2)
        u1c = np.zeros(1024 * 3 // 2 + 64, np.int8)
        u1d = np.zeros(1024 + 64, np.uint8)
        dfn8, cfn8 = _lib.decompress[("delta", 1)], _lib.compress[("delta", 1)]
        for _ in range(20):
            cfn8(u1.ctypes.data, 1024, u1c.ctypes.data, 1, 1)
            dfn8(u1c.ctypes.data, u1d.ctypes.data)
        l8d, l8c = [], []
        for _ in range(300):
            t0 = time.perf_counter(); cfn8(u1.ctypes.data, 1024, u1c.ctypes.data, 1, 1); l8c.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); r = dfn8(u1c.ctypes.data, u1d.ctypes.data); l8d.append(time.perf_counter() - t0)
            assert r == 1024
        assert np.array_equal(u1d[:1024], u1)
        l8d.sort(); l8c.sort()
        res["single_call_latency_cfg1_1KB"] = {"decompress_us_median": round(l8d[150] * 1e6, 1), "compress_us_median": round(l8c[150] * 1e6, 1),
                                               "what": "sprintz_{de,}compress_delta_8b on one 1 KB univariate chunk (BASELINE configs[0]'s shape; low-dim layout, "
                                                       "the same one-workgroup-per-chunk kernels)"}
        # ---------------- the same drop-in symbols from many host threads at once (each thread has its own pooled scratch and stream
        # inside the library; ctypes releases the interpreter lock around the call): calls per second of the whole process
        def many_threads(nthreads, calls):
            import threading
            bufs = [(np.zeros(chunk_len + 64, np.uint16), np.zeros(chunk_len * 3 // 2 + 64, np.int16)) for _ in range(nthreads)]
            go = threading.Barrier(nthreads + 1)

            def work(k):
                d, c = bufs[k]
                dfn(one.ctypes.data, d.ctypes.data)                       # first call of the thread: scratch, stream
                go.wait()
                for _ in range(calls):
                    dfn(one.ctypes.data, d.ctypes.data)
                    cfn(raw1.ctypes.data, chunk_len, c.ctypes.data, ndims, 1)
            ths = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
            [t.start() for t in ths]
            go.wait()
            t0 = time.perf_counter()
            [t.join() for t in ths]
            dt = time.perf_counter() - t0
            ok = all(np.array_equal(b[0][:chunk_len], raw1) for b in bufs)
            return 2 * nthreads * calls / dt, ok
        res["single_call_threads"] = {}
        for nt in (1, 8, 64):
            rate, ok = many_threads(nt, 200 if nt < 64 else 60)
            assert ok, "a thread's single-call decode differs from the input"
            res["single_call_threads"][str(nt)] = {"calls_per_s": round(rate), "MBps_of_samples": round(rate * chunk_bytes / 1e6, 1)}
        res["single_call_threads"]["what"] = ("N PYTHON threads (ctypes releases the interpreter lock around a call, not between calls), each alternating "
                                              "sprintz_decompress_xff_16b / sprintz_compress_xff_16b on its own 10 KB chunk; 'native' = the same loop from N "
                                              "pthreads (tools/probes/single_call_mt.cpp, a subprocess): what a multi-threaded lzbench-style driver gets out "
                                              "of the single-call boundary (the threads share 4 streams per device, SPRINTZ_OPT_HOST_STREAMS; up to 4 callers "
                                              "spin on their call's flag word, more sleep and poll it); the batched device API is the fast path")
        res["single_call_threads"]["native"] = native_threads_leg()
        # ---------------- online.hpp's u16 coders: ONE stream of 64 Mi samples per call, device buffers
        res["online_coders"] = online_leg(cx)
        # ---------------- the stand-alone transforms (delta.cpp, predict.cpp) and a 1 000-column shape (csrc/any_ndims.hip)
        try:
            res["transforms"] = transforms_leg(cx)
            res["any_ndims"] = any_ndims_leg(cx)
        except Exception as e:      # noqa: BLE001 -- a failing leg must not cost the headline line
            res["transforms"] = {"failed": repr(e)[:200]}
        # ---------------- real (measured) data: what the image holds without a network
        try:
            res["real_data"] = real_data_leg(cx)
        except Exception as e:      # noqa: BLE001 -- a failing leg must not cost the headline line
            res["real_data"] = {"failed": repr(e)[:200]}
        # ---------------- PCIe-inclusive: host buffers in and out through the chunked host entry points
        ns = min(nchunks, 16384)
        oh = offsets[: ns + 1].cpu().numpy().astype(np.uint64)
        ch = np.ascontiguousarray(comp[: int(oh[ns]) + 64].cpu().numpy())
        outh = np.zeros(ns * chunk_len, np.uint16)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            r = _lib.decompress_chunked_host(_lib.CODEC_XFF, esz, ch.ctypes.data, oh.ctypes.data, ns, chunk_len, ndims, outh.ctypes.data)
            d = time.perf_counter() - t0
            assert r == ns * chunk_len
            best = d if best is None else min(best, d)
        res["pcie_inclusive"] = {"decompress_MBps": round(ns * chunk_bytes / best / 1e6, 1), "chunks": ns,
                                 "what": "sprintz_mi355x_decompress_chunked_host: pageable host buffers in and out (H2D of the streams, kernel, D2H "
                                         "of the samples, allocations included); bounded by the 63 GB/s PCIe Gen5 link -- never `value`"}
    return res


def transforms_leg(cx):
    """delta / double delta / FIRE as stand-alone transforms (delta.cpp:35,533; predict.cpp:57,302) on ONE stream of 64 Mi uint16 samples x 8
    columns' worth of rows: bytes in = bytes out, so frac = 2 x stream bytes / time over 8 TB/s.  -> {kind: [enc_ms, enc_frac, dec_ms, dec_frac]}"""
    torch, dev, timed = cx.torch, cx.device, cx.timer
    import sprintz_amd
    from synth import synth_torch
    D, rows = 8, (64 << 20) // 8
    x = synth_torch("walk", 2, 1, rows, D, dev, seed=123, step=8).reshape(-1)
    nbytes = x.numel() * 2
    out = {"samples": x.numel(), "ndims": D, "fields": "enc_ms, enc_frac, dec_ms, dec_frac"}
    for kind in ("delta", "doubledelta"):
        y, back = torch.empty_like(x), torch.empty_like(x)
        te = timed(lambda: sprintz_amd.transform_device(kind, x, D, out=y), 10, 2)
        td = timed(lambda: sprintz_amd.transform_device(kind, y, D, inverse=True, out=back), 10, 2)
        assert torch.equal(back.view(torch.int16), x.view(torch.int16)), kind
        out[kind] = [round(te, 4), round(2 * nbytes / (te * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), round(td, 4), round(2 * nbytes / (td * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)]
    # the same two on a stream eight times as long (1 GiB): at 128 MiB a pass is 40 - 130 us and the decode's three launches are a sixth of it
    xl = x.repeat(8)
    big = {}
    for kind in ("delta", "doubledelta"):
        y, back = torch.empty_like(xl), torch.empty_like(xl)
        te = timed(lambda: sprintz_amd.transform_device(kind, xl, D, out=y), 4, 1)
        td = timed(lambda: sprintz_amd.transform_device(kind, y, D, inverse=True, out=back), 4, 1)
        assert torch.equal(back.view(torch.int16), xl.view(torch.int16)), kind
        big[kind] = [round(te, 4), round(16 * nbytes / (te * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), round(td, 4), round(16 * nbytes / (td * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)]
        del y, back
    out["at_512Mi_samples"] = big
    del xl
    # FIRE as a transform of ONE stream is a recurrence down every column over the WHOLE stream (the counters never reset): eight lanes of
    # work for eight columns, however long the stream -- a latency, measured on a sixteenth of the stream and labelled as such
    xs = x[: x.numel() // 16]
    y, back = torch.empty_like(xs), torch.empty_like(xs)
    te = timed(lambda: sprintz_amd.transform_device("xff", xs, D, out=y), 1, 1)
    td = timed(lambda: sprintz_amd.transform_device("xff", y, D, inverse=True, out=back), 1, 1)
    assert torch.equal(back.view(torch.int16), xs.view(torch.int16)), "xff"
    out["xff_serial_4Mi_samples"] = [round(te, 3), round(4 * xs.numel() / (te * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), round(td, 3), round(4 * xs.numel() / (td * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)]
    return out


def any_ndims_leg(cx):
    """uint16 x 1 000 columns, FIRE, chunks of 16 groups (256 rows = 512 000 bytes), 1 024 chunks = 524 MB: the one-workgroup-per-chunk kernels
    of csrc/any_ndims.hip (513 .. 65 535 columns).  -> {dec_ms, dec_frac, enc_ms, enc_frac, ratio}"""
    torch, dev, timed = cx.torch, cx.device, cx.timer
    import sprintz_amd
    from synth import synth_torch
    D, rows, n = 1000, 256, 1024
    x = synth_torch("walk", 2, n, rows, D, dev, seed=123, step=8).view(torch.int16)
    cd = sprintz_amd.ChunkedCodec("xff", 2, D, rows * D, device=dev)
    batch = cd.compress(x)
    out = torch.empty_like(x)
    te = timed(lambda: cd.compress(x), 5, 1)
    td = timed(lambda: cd.decompress(batch, out=out), 5, 1)
    assert torch.equal(out, x), "any_ndims: GPU decode != input"
    raw, sb = x.numel() * 2, batch.stream_bytes()
    return {"workload": "uint16 x 1000 columns, FIRE, 256-row chunks, 1024 chunks", "ratio": round(raw / sb, 4),
            "dec_ms": round(td, 4), "dec_frac": round((raw + sb) / (td * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "enc_ms": round(te, 4), "enc_frac": round((raw + sb) / (te * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def online_leg(cx):
    """dynamic delta / zigzag / sprintzpack over one 128 MiB uint16 walk: GB/s of samples, pack and unpack, round trip checked; and the fractions
    again on a stream eight times as long (1 GiB: `frac_at_512Mi_samples` = [pack, unpack])"""
    torch, dev, timed = cx.torch, cx.device, cx.timer
    from sprintz_amd import _lib
    from synth import synth_torch
    n = 64 << 20
    x = synth_torch("walk", 2, 1, n, 1, dev, seed=123, step=8)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {"samples": n, "what": "sprintz_mi355x_online_{pack,unpack}_device on one uint16 stream (walk, steps in [-8, 8]); *_frac: (samples + container bytes) / time over 8 TB/s"}
    big = {}
    nl = 8 * n
    xl = x.reshape(-1).repeat(8)
    for name, kind in (("dynamic_delta", 0), ("zigzag", 2), ("sprintzpack", 3)):
        dest = torch.zeros(int(_lib.online_bound(kind, nl)) + 64, dtype=torch.uint8, device=dev)
        tmp = torch.zeros(int(_lib.online_tmp_bytes(kind, nl)) + 64, dtype=torch.uint8, device=dev)
        back = torch.empty(nl, dtype=torch.uint16, device=dev)
        ret = torch.zeros(2, dtype=torch.int64, device=dev)
        p_ms = timed(lambda: _lib.check(_lib.online_pack_device(kind, xl.data_ptr(), nl, dest.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st)), 3, 1)
        elems = int(ret[0].item())
        u_ms = timed(lambda: _lib.check(_lib.online_unpack_device(kind, dest.data_ptr(), nl, back.data_ptr(), ret.data_ptr() + 8, tmp.data_ptr(), st)), 3, 1)
        assert int(ret[1].item()) == nl and torch.equal(back.view(torch.int16), xl.view(torch.int16)), name
        moved = 2 * nl + 2 * elems
        big[name] = [round(moved / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), round(moved / (u_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)]
        del dest, tmp, back
    out["frac_at_512Mi_samples"] = big
    del xl
    for name, kind in (("dynamic_delta", 0), ("zigzag", 2), ("sprintzpack", 3), ("sprintzpack_zigzag", 4)):
        dest = torch.zeros(int(_lib.online_bound(kind, n)) + 64, dtype=torch.uint8, device=dev)
        tmp = torch.zeros(int(_lib.online_tmp_bytes(kind, n)) + 64, dtype=torch.uint8, device=dev)
        back = torch.empty(n, dtype=torch.uint16, device=dev)
        ret = torch.zeros(2, dtype=torch.int64, device=dev)

        def pack():
            _lib.check(_lib.online_pack_device(kind, x.data_ptr(), n, dest.data_ptr(), ret.data_ptr(), tmp.data_ptr(), st))

        def unpack():
            _lib.check(_lib.online_unpack_device(kind, dest.data_ptr(), n, back.data_ptr(), ret.data_ptr() + 8, tmp.data_ptr(), st))
        p_ms = timed(pack, 5, 1)
        elems = int(ret[0].item())
        u_ms = timed(unpack, 5, 1)
        assert int(ret[1].item()) == n and torch.equal(back.view(torch.int16), x.view(torch.int16)), name
        moved = 2 * n + 2 * elems                      # algorithmic bytes either way: the samples + the container
        out[name] = {"ratio": round(n / max(elems, 1), 4), "pack_ms": round(p_ms, 3), "pack_GBps": round(2 * n / p_ms / 1e6, 1),
                     "unpack_ms": round(u_ms, 3), "unpack_GBps": round(2 * n / u_ms / 1e6, 1),
                     "pack_frac": round(moved / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "unpack_frac": round(moved / (u_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return out


def real_data_leg(cx):
    """the tabular sets and photographs bundled with scikit-learn (sprintz_amd.datasets.offline_real_datasets), quantised to 8 and 16 bits per
    variable as python/datasets/compress_bench.py:45-60 does, FIRE codec, 10 KB chunks: ratio per set (round trip checked).  They are
    small -- 42 MB altogether would be a luxury; these are 6 MB -- so this leg reports ratios, not throughput."""
    import numpy as np
    torch, dev = cx.torch, cx.device
    import sprintz_amd
    from sprintz_amd import datasets
    sets = datasets.offline_real_datasets()
    if not sets:
        return {"skipped": "scikit-learn's bundled datasets are not importable"}
    out = {"what": "scikit-learn's bundled tabular sets + photographs (pixel by pixel: 3 variables; scan line by scan line: 1 920), quantised per "
                   "variable as compress_bench.py:45-60, sprintz_xff, chunks of ~10 KB (at least 64 rows), decode checked; [ratio at 8 bits, ratio at 16 bits]",
           "sets": {}}
    raw_total = comp_total = 0
    for name, mat in sets:
        pair = []
        for dt in (np.uint8, np.uint16):
            q = datasets.quantize(mat, dt)
            esz, ndims = q.dtype.itemsize, q.shape[1]
            rows = min(max(64, (10240 // (ndims * esz)) // 8 * 8), max(8, q.shape[0] // 8 * 8))      # ~10 KB of rows, at least 64
            cd = sprintz_amd.ChunkedCodec("xff", esz, ndims, rows * ndims, device=dev)
            t = torch.from_numpy(np.ascontiguousarray(q).view(np.int8 if esz == 1 else np.int16)).to(dev).view(cd.dtype)
            b = cd.compress(t)
            back = cd.decompress(b)
            assert torch.equal(back.view(torch.uint8)[: q.nbytes], t.view(torch.uint8).reshape(-1)), (name, str(dt))
            pair.append(round(q.nbytes / b.total_bytes(), 3))
            raw_total += q.nbytes
            comp_total += b.total_bytes()
        out["sets"][name] = pair
    out["raw_bytes"] = raw_total
    out["ratio_overall"] = round(raw_total / comp_total, 3)
    return out


if __name__ == "__main__":
    main()
