"""Dataset plumbing: what the reference's
python/datasets/compress_bench.py:45-120 does to feed real data to the codec -- quantise a float
matrix to uint8/uint16 per variable and dump it row-major ("all variables of a timestamp
contiguous") or column-major ("all samples of a variable contiguous") as raw little-endian
.dat files -- plus the way back in: load such a file and hand it to ChunkedCodec in the
layout it is stored in (column-major files go through compress_colmajor untransposed).
Host-side numpy; nothing here is on the GPU path."""
import numpy as np

_MAX = {np.dtype(np.uint8): 255, np.dtype(np.uint16): 65535}


def quantize(mat, dtype, axis=0):
    """compress_bench.py:45-60: shift every variable (column) to start at 0, scale it to the full
    range of dtype (a constant column stays 0; the reference divides by max(1, column max)),
    truncate.  mat: [nsamples, nvariables]."""
    dt = np.dtype(dtype)
    if dt not in _MAX:
        raise ValueError(f"Invalid dtype '{dtype}'")
    m = np.array(mat, dtype=np.float64, copy=True)
    if m.ndim == 1:
        m = m[:, None]
    m -= np.min(m, axis=axis, keepdims=True)
    m = m.astype(np.float32)
    m /= np.maximum(1, np.max(m, axis=axis, keepdims=True))
    return (m * _MAX[dt]).astype(dt)


def dump(mat, path, order="c"):
    """write [nsamples, nvariables] as a raw little-endian .dat: order 'c' = row-major, 'f' =
    column-major (compress_bench.py:111-115)"""
    a = np.ascontiguousarray(mat if order == "c" else np.asarray(mat).T)
    a.astype(a.dtype.newbyteorder("<"), copy=False).tofile(path)
    return path


def load(path, dtype, ndims, order="c"):
    """-> array in the layout of the file: [nsamples, ndims] for 'c', [ndims, nsamples] for 'f'"""
    flat = np.fromfile(path, dtype=np.dtype(dtype).newbyteorder("<"))
    if flat.size % ndims:
        raise ValueError(f"{path}: {flat.size} elements is not a multiple of ndims={ndims}")
    return flat.reshape(-1, ndims) if order == "c" else flat.reshape(ndims, -1)


def compress_file(path, dtype, ndims, order="c", codec="xff", rows_per_chunk=None, device=None):
    """load a .dat dump and compress it on the GPU in the layout it is stored in.
    -> (ChunkedCodec, CompressedBatch).  rows_per_chunk defaults to 10 KB worth of rows."""
    import torch
    from .codec import ChunkedCodec
    a = load(path, dtype, ndims, order)
    esz = np.dtype(dtype).itemsize
    if rows_per_chunk is None:
        rows_per_chunk = max(8, (10240 // (ndims * esz)) // 8 * 8)
    cd = ChunkedCodec(codec, esz, ndims, rows_per_chunk * ndims, device=device)
    t = torch.from_numpy(a.view(np.int8 if esz == 1 else np.int16)).to(cd.device).view(cd.dtype)
    return cd, (cd.compress(t) if order == "c" else cd.compress_colmajor(t))


def offline_real_datasets():
    """Real (measured, not generated) data that is present without a network: the tabular sets and the two photographs that ship
    inside scikit-learn.  -> [(name, float64 matrix [nsamples, nvariables])]; empty if scikit-learn (or Pillow, for the
    photographs) is not importable.  None of them is a time series -- the paper's archives (UCR, AMPds, MSRC-12, PAMAP, UCI gas:
    compress_bench.py:80-90 feeds those) are not in the image -- but they are what real quantised measurements look like to the
    codec: correlated neighbours, plateaus, heavy tails, constant columns.  The photographs are taken twice: pixel by pixel
    (3 variables: R, G, B down the image) and scan line by scan line (1 920 variables: a row of 640 RGB pixels per sample)."""
    out = []
    try:
        from sklearn import datasets as skd
    except Exception:                                        # noqa: BLE001 -- no scikit-learn: no real data, the caller says so
        return out
    for name in ("load_diabetes", "load_breast_cancer", "load_wine", "load_digits", "load_iris", "load_linnerud"):
        try:
            out.append((name[5:], np.asarray(getattr(skd, name)().data, dtype=np.float64)))
        except Exception:                                    # noqa: BLE001
            pass
    try:
        images = skd.load_sample_images()
        for fname, im in zip(images.filenames, images.images):
            base = str(fname).replace("\\", "/").rsplit("/", 1)[-1].rsplit(".", 1)[0]
            a = np.asarray(im, dtype=np.float64)
            out.append((base + "_pixels", a.reshape(-1, a.shape[-1])))
            out.append((base + "_scanlines", a.reshape(a.shape[0], -1)))
    except Exception:                                        # noqa: BLE001 -- no Pillow
        pass
    return out
