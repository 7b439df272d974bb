"""The oracle under UBSan (oracle/Makefile: liboracle_asan.so = every oracle .c file with -fsanitize=undefined
-fno-sanitize-recover=all): the golden sets go through the sanitised build in a child process -- any undefined behaviour in
the restatement (a shift past the width, a signed overflow that the scalar arithmetic relies on, a misaligned access) aborts
the child, and the bytes must still be the golden ones.  CPU only; the product never links this library."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from harness import Oracle
orc = Oracle(os.path.join(sys.argv[1], "oracle", "liboracle_asan.so"))
gdir = os.path.join(sys.argv[1], "tests", "golden")
man = json.load(open(os.path.join(gdir, "golden_v1.json")))["cases"]
arr = np.load(os.path.join(gdir, "golden_v1.npz"))
n = 0
for m in man:
    data, want = arr[f"in_{m['idx']}"], arr[f"out_{m['idx']}"]
    got, ret = orc.compress(m["codec"], data, m["ndims"])
    assert ret == m["ret"] and np.array_equal(got, want), m
    dec, dret = orc.decompress(m["codec"], want, m["esz"], data.size)
    assert dret == data.size and np.array_equal(dec, data.ravel()), m
    n += 1
# the Huff0 reader on libzstd's blocks
man = json.load(open(os.path.join(gdir, "golden_huf0_v1.json")))["cases"]
arr = np.load(os.path.join(gdir, "golden_huf0_v1.npz"))
h = 0
for m in man:
    plain_want, block = arr["p%04d" % m["idx"]], arr["b%04d" % m["idx"]]
    if block.size >= plain_want.size or block.size <= 1:        # stored / single-symbol cases: HUF_decompress's other branches
        continue
    plain, ret = orc.huf0_decompress(block, plain_want.size)
    assert ret == plain_want.size and np.array_equal(plain, plain_want), m
    h += 1
print("SANITIZED_OK", n, h)
"""


def test_oracle_under_ubsan_reproduces_the_golden_vectors():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "runtime error" not in p.stderr, p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("SANITIZED_OK")]
    assert line and int(line[0].split()[1]) >= 600 and int(line[0].split()[2]) >= 200, p.stdout[-500:]
