import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# The two-column encoder (encode_wide.h) takes row-major batches from 1 024 chunks on by default (SPRINTZ_OPT_ENC_PAIR); most tests are
# smaller than that, so the test session asks for it from one chunk on -- the tests that compare it with the one-column encoder
# (encode_fast.h) switch the option themselves.
os.environ.setdefault("SPRINTZ_MI355X_ENC_PAIR", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(params=["lat", "wide", "blk"])
def decode_path(request):
    """the three families of row-major kernels: small batches (by default up to 2 048 chunks) take decode_lat.h / encode_lat.h (one
    workgroup per chunk), larger ones decode_fast.h / decode_kernel.h / encode_wide.h (one lane per column or column pair), large
    batches of the DELTA codec the block-parallel kernels (encode_blk.h, decode_blk.h; SPRINTZ_OPT_BLK_CHUNKS).  The parity modules run
    every test on all three: "lat" sends every eligible batch to the workgroup-per-chunk kernels whatever its size, "wide" switches both
    other families off, "blk" sends every eligible batch -- one chunk included -- to the block-parallel kernels."""
    from sprintz_amd import _lib
    _lib.check(_lib.set_option(_lib.OPT_LAT_CHUNKS, (1 << 30) if request.param == "lat" else 0))
    _lib.check(_lib.set_option(_lib.OPT_BLK_CHUNKS, 1 if request.param == "blk" else 0))
    _lib.check(_lib.set_option(_lib.OPT_BLK_KERNELS, 31))           # every new kernel, whatever the default mask is (the decoder: decode_row.h; decode_blk.h has tests/test_gpu_blk.py)
    yield request.param
    _lib.set_option(_lib.OPT_LAT_CHUNKS, int(os.environ.get("SPRINTZ_MI355X_LAT_CHUNKS", 2048)))
    _lib.set_option(_lib.OPT_BLK_CHUNKS, int(os.environ.get("SPRINTZ_MI355X_BLK_CHUNKS", 2049)))
    _lib.set_option(_lib.OPT_BLK_KERNELS, int(os.environ.get("SPRINTZ_MI355X_BLK_KERNELS", 9)))


@pytest.fixture(scope="session")
def oracle():
    """Our CPU restatement; built on demand (gcc only, a second or two)."""
    import subprocess
    from harness import ORACLE_SO, Oracle
    # always through make: a no-op when liboracle.so is current, a rebuild when the sources moved on
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference, if oracle/_ref was built (needs /root/reference at build time)."""
    from harness import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libsprintz_ref.so not built (reference sources absent)")
    return Reference()


@pytest.fixture(scope="session")
def golden_rowmajor():
    """streams of the reference's *_rowmajor_*_rle_* family (oracle/gen_golden_rowmajor.py)"""
    import json
    import numpy as np
    gdir = os.path.join(HERE, "golden")
    with open(os.path.join(gdir, "golden_rowmajor_v1.json")) as f:
        manifest = json.load(f)["cases"]
    arrays = np.load(os.path.join(gdir, "golden_rowmajor_v1.npz"))
    return manifest, arrays


@pytest.fixture(scope="session")
def golden_transforms():
    """containers of the reference's stand-alone transforms (oracle/gen_golden_transforms.py)"""
    import json
    import numpy as np
    gdir = os.path.join(HERE, "golden")
    with open(os.path.join(gdir, "golden_transforms_v2.json")) as f:
        manifest = json.load(f)["cases"]
    arrays = np.load(os.path.join(gdir, "golden_transforms_v2.npz"))
    return manifest, arrays


@pytest.fixture(scope="session")
def golden_huf0():
    """Huff0 blocks written by libzstd's HUF_compress (oracle/gen_golden_huf0.py)"""
    import json
    import numpy as np
    gdir = os.path.join(HERE, "golden")
    with open(os.path.join(gdir, "golden_huf0_v1.json")) as f:
        manifest = json.load(f)["cases"]
    arrays = np.load(os.path.join(gdir, "golden_huf0_v1.npz"))
    return manifest, arrays


@pytest.fixture(scope="session")
def golden_norle():
    """streams of the reference's non-RLE codecs (oracle/gen_golden_norle.py)"""
    import json
    import numpy as np
    gdir = os.path.join(HERE, "golden")
    with open(os.path.join(gdir, "golden_norle_v1.json")) as f:
        manifest = json.load(f)["cases"]
    arrays = np.load(os.path.join(gdir, "golden_norle_v1.npz"))
    return manifest, arrays


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    gdir = os.path.join(HERE, "golden")
    with open(os.path.join(gdir, "golden_v1.json")) as f:
        manifest = json.load(f)["cases"]
    arrays = np.load(os.path.join(gdir, "golden_v1.npz"))
    return manifest, arrays


@pytest.fixture(scope="session")
def golden_wide2():
    """the reference's streams at 2 048 .. 65 535 columns (oracle/gen_golden_wide.py v2): -> (manifest, arrays); a case's input is arrays["in_%d" % m["in_idx"]]"""
    import json
    import numpy as np
    gdir = os.path.join(HERE, "golden")
    with open(os.path.join(gdir, "golden_wide_v2.json")) as f:
        manifest = json.load(f)["cases"]
    arrays = np.load(os.path.join(gdir, "golden_wide_v2.npz"))
    return manifest, arrays


@pytest.fixture(scope="session")
def golden_wide():
    """the reference's streams at 513 .. 2 047 columns (oracle/gen_golden_wide.py)"""
    import json
    import numpy as np
    gdir = os.path.join(HERE, "golden")
    with open(os.path.join(gdir, "golden_wide_v1.json")) as f:
        manifest = json.load(f)["cases"]
    arrays = np.load(os.path.join(gdir, "golden_wide_v1.npz"))
    return manifest, arrays
