"""bench.py's launch path and its many-core CPU baseline, without a GPU.

* `python bench.py --gpus N` with no launcher around it must bring its N ranks up itself (round 2 asserted
  WORLD_SIZE == N and died): `--dry-launch` runs exactly that path -- self-launch under torch.distributed.run, process
  group, an all-gather of the ranks, ONE JSON line from rank 0 -- with gloo when there is no GPU.
* the all-core leg of `cpu_baseline` runs in C (oracle/mt_bench.c: pinned pthreads, one chunk range each): it must decode
  what the single-threaded chunk loop decodes.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, BENCH] + argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return p


@pytest.mark.parametrize("n", [2, 4, 8])
def test_gpus_n_launches_its_own_ranks(n):
    p = run_bench(["--gpus", str(n), "--dry-launch"], {"BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                     # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["self_launched"] is True and d["world"] == n and d["n_gpus"] == n
    assert d["ranks"] == list(range(n))


def test_one_gpu_needs_no_launcher():
    p = run_bench(["--dry-launch"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["world"] == 1 and d["self_launched"] is False


def test_a_launcher_with_the_wrong_world_size_is_named():
    p = run_bench(["--gpus", "4", "--dry-launch"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0
    assert "WORLD_SIZE=2" in p.stderr and "--nproc-per-node 4" in p.stderr


def test_many_core_baseline_decodes_what_one_thread_decodes(oracle):
    """oracle_mt_decompress_chunks over the oracle's own chunk loop: every thread count gives the single-thread samples"""
    from harness import gen_walk
    lib = oracle.lib
    ndims, chunk_len, nchunks = 8, 5120, 37
    rng = np.random.default_rng(5)
    data = gen_walk(rng, nchunks * chunk_len, ndims, 2, 8, flat_every=4)
    streams = oracle.compress_chunks("xff", data, chunk_len, ndims)
    offs = np.zeros(nchunks + 1, np.uint64)
    offs[1:] = np.cumsum([s.size for s in streams])
    comp = np.concatenate(streams + [np.zeros(64, np.uint8)])
    mt = lib.oracle_mt_decompress_chunks
    mt.restype = C.c_double
    mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                   C.c_void_p, C.c_void_p]
    fn = C.cast(lib.oracle_decompress_chunks, C.c_void_p)
    gap = 256
    for nthreads, pin in ((1, False), (3, False), (8, True), (64, False)):
        out = np.zeros(nchunks * chunk_len * 2 + 65 * gap, np.uint8)
        elems = C.c_uint64(0)
        cpus = None
        if pin:
            avail = sorted(os.sched_getaffinity(0))
            cpus = (C.c_int * nthreads)(*[avail[i % len(avail)] for i in range(nthreads)])
        t = mt(fn, 1, 2, comp.ctypes.data, offs.ctypes.data, nchunks, chunk_len, out.ctypes.data, gap, nthreads, 2, cpus, C.byref(elems))
        assert t > 0 and elems.value == nchunks * chunk_len
        nt = min(nthreads, nchunks)
        for th in range(nt):
            lo, hi = nchunks * th // nt, nchunks * (th + 1) // nt
            got = out[lo * chunk_len * 2 + th * gap: hi * chunk_len * 2 + th * gap].view(np.uint16)
            assert np.array_equal(got, data[lo * chunk_len: hi * chunk_len]), (nthreads, th)


def test_host_topology_counts_physical_cores():
    sys.path.insert(0, ROOT)
    import bench
    avail, firsts = bench.host_topology()
    assert 1 <= len(firsts) <= len(avail)
    assert set(firsts) <= set(avail) and len(set(firsts)) == len(firsts)
