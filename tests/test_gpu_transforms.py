"""GPU tests (-m gpu): stand-alone transforms (delta.h:17-68, predict.h:15-30) through the C-ABI, against the
containers minted from the compiled reference, the oracle, and at sizes where the decode's
multi-level scan has several levels.  Nothing here reads /root/reference."""
import numpy as np
import pytest

from harness import DTYPES

pytestmark = pytest.mark.gpu

NAMES = {0: "delta", 1: "doubledelta", 2: "xff"}


@pytest.fixture(scope="module")
def sz():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import sprintz_amd
    return sprintz_amd


def test_reference_containers_single_call(sz, golden_transforms):
    """encode_* writes the reference's bytes and return value; decode_* (both forms) restores the input"""
    manifest, arrays = golden_transforms
    for m in manifest:
        x, cont = arrays[m["name"] + "_in"], arrays[m["name"] + "_container"]
        esz, D, n, kind = m["esz"], m["ndims"], m["n"], m["kind"]
        enc = getattr(sz, f"encode_{NAMES[kind]}_rowmajor_{8 * esz}b")
        dec = getattr(sz, f"decode_{NAMES[kind]}_rowmajor_{8 * esz}b")
        dest = np.full(cont.size + 32, 0xAB, np.uint8)
        ret = enc(x, n, dest, D)
        assert ret == m["ret"], m
        assert np.array_equal(dest[:cont.size], cont) and (dest[cont.size:] == 0xAB).all(), m
        out = np.full(n + 16, 0xCD, DTYPES[esz])
        assert dec(cont, out) == n and np.array_equal(out[:n], x) and (out[n:] == 0xCD).all(), m
        out[:] = 0xCD
        assert dec(cont[6:], out, n, D) == n and np.array_equal(out[:n], x), m       # headerless 4-argument form
        dest2 = np.full(cont.size, 0xAB, np.uint8)
        assert enc(x, n, dest2, D, False) == n and np.array_equal(dest2[:n * esz], cont[6:]), m   # write_size=false


@pytest.mark.parametrize("kind", ["delta", "doubledelta"])
@pytest.mark.parametrize("esz,ndims,n", [(2, 8, 8 * 1_000_003), (1, 80, 80 * 70_001 + 13), (2, 3, 3 * 5_000_000 + 1), (1, 1, 3_000_001),
                                         (2, 300, 300 * 40_000), (2, 32, 32 * (1 << 20)),
                                         # ragged last rows: the wave-scan path drops to 4-byte and 2-/1-byte pieces
                                         (2, 8, 8 * 100_003 + 4), (2, 8, 8 * 100_003 + 3), (1, 16, 16 * 50_001 + 7), (2, 2, 2 * 70_001 + 1),
                                         # rows shorter than a 16-byte piece (several rows per piece, in-register prefix)
                                         (1, 1, 1 << 24), (1, 1, 16 * 4097), (1, 2, 16 * 300_001), (1, 4, 4 * 1_000_004), (1, 8, 8 * 500_002),
                                         (2, 1, 8 * 700_001), (2, 2, 2 * 4_000_004), (2, 4, 4 * 600_002), (1, 1, 16), (2, 1, 65536 * 8 + 8),
                                         # piece counts that are not powers of two: 5, 3, 9, 37 pieces a row
                                         (1, 80, 80 * 300_001), (2, 24, 24 * 200_003), (1, 144, 144 * 70_001), (2, 296, 296 * 30_011), (2, 40, 40 * 17)])
@pytest.mark.parametrize("chain", ["default", "1", "0"])
def test_long_streams_on_device(sz, oracle, kind, esz, ndims, n, chain, monkeypatch):
    """one stream of millions of rows: element-wise encode, scan decode -- the one-pass chained scan over tiles (default: every 16-bit stream; "1":
    from the first tile on, so that the short streams take it too) and the two-pass form with its 3 to 6 levels ("0")"""
    import torch
    if chain != "default":
        monkeypatch.setenv("SPRINTZ_MI355X_TRANSFORM_CHAIN", chain)
    g = torch.Generator(device="cuda")
    g.manual_seed(n % 1000)
    x = torch.randint(0, 1 << (8 * esz), (n,), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8 if esz == 1 else torch.uint16)
    y = sz.transform_device(kind, x, ndims)
    # spot-check the encode against the oracle on the head and against the definition on the whole
    head = x[: 64 * ndims + 5].cpu().numpy().view(DTYPES[esz])
    cont, _ = oracle.transform_encode(0 if kind == "delta" else 1, head, ndims)
    assert np.array_equal(y[: head.size].cpu().numpy().view(DTYPES[esz]), cont[6:].view(DTYPES[esz]))
    xi = x.view(torch.int8 if esz == 1 else torch.int16).to(torch.int32)
    p1 = torch.zeros_like(xi); p1[ndims:] = xi[:-ndims]
    want = xi - p1
    if kind == "doubledelta":
        p2 = torch.zeros_like(xi); p2[2 * ndims:] = xi[:-2 * ndims]
        want = xi - 2 * p1 + p2
    mask = (1 << (8 * esz)) - 1
    assert torch.equal(y.view(torch.int8 if esz == 1 else torch.int16).to(torch.int32) & mask, want & mask)
    back = sz.transform_device(kind, y, ndims, inverse=True)
    assert torch.equal(back, x)


@pytest.mark.parametrize("esz,ndims,n", [(2, 8, 8 * 200_003 + 5), (1, 80, 80 * 20_001 + 13), (1, 1, 100_001), (2, 300, 300 * 4_000 + 7),
                                         (1, 33, 33 * 10_000), (2, 16, 16 * 8 * 5000)])
def test_xff_long_streams_on_device(sz, oracle, esz, ndims, n):
    """FIRE errors of one long stream (predict.cpp): device forms against the oracle, and back"""
    import torch
    rng = np.random.default_rng(n % 997)
    top = 1 << (8 * esz)
    x = ((np.cumsum(rng.integers(-6, 7, n)) + rng.integers(0, 3, n)) % top).astype(DTYPES[esz])
    x[n // 3: n // 3 + 5000] = rng.integers(0, top, 5000).astype(DTYPES[esz])        # a noisy stretch: counters swing
    want, _ = oracle.transform_encode(2, x, ndims)
    xd = torch.from_numpy(x.view(np.uint8 if esz == 1 else np.int16)).cuda()
    xd = xd if esz == 1 else xd.view(torch.uint16)
    y = sz.transform_device("xff", xd, ndims)
    assert np.array_equal(y.cpu().numpy().view(DTYPES[esz]), want[6:].view(DTYPES[esz]))
    back = sz.transform_device("xff", y, ndims, inverse=True)
    assert torch.equal(back, xd)


def test_one_pass_decoders_on_two_streams_at_once(sz):
    """the one-pass decoders' workgroups wait for each other (a chained scan over tiles, tiles taken by ticket): two decodes in flight at once on
    two streams -- neither launch has the chip to itself, a waiting tile's predecessors are still held by running workgroups -- must both finish"""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    n = 8 * (4 << 20)                                                   # 64 MB of uint16 x 8 columns: 512 tiles, two rounds of the persistent grid
    xs = [torch.randint(0, 1 << 16, (n,), generator=g, device="cuda", dtype=torch.int32).to(torch.uint16) for _ in range(2)]
    ys = [sz.transform_device("delta", xs[0], 8), sz.transform_device("doubledelta", xs[1], 8)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    outs = [[], []]
    for it in range(6):
        for k, kind in enumerate(("delta", "doubledelta")):
            with torch.cuda.stream(streams[k]):
                outs[k].append(sz.transform_device(kind, ys[k], 8, inverse=True))
    torch.cuda.synchronize()
    for k in range(2):
        for o in outs[k]:
            assert torch.equal(o, xs[k])
