"""Independent NIfTI-1 reader / writer for the tests (struct + gzip, nothing shared with the C++
implementation under test). Volumes are numpy arrays indexed [x, y, z] or [x, y, z, t]."""
import gzip
import struct

import numpy as np

DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32}
CODES = {np.dtype(v): k for k, v in DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if path.endswith(".gz") else open(path, mode)


def write(path, vol, pixdim=(1.0, 1.0, 1.0, 1.0), intent_code=0, byteorder="<", scl=(0.0, 0.0)):
    vol = np.asarray(vol)
    if vol.ndim == 3:
        vol = vol[..., None]
    nx, ny, nz, nt = vol.shape
    code = CODES[vol.dtype]
    e = byteorder
    hdr = bytearray(348)
    struct.pack_into(e + "i", hdr, 0, 348)
    struct.pack_into(e + "8h", hdr, 40, 4 if nt > 1 else 3, nx, ny, nz, nt, 1, 1, 1)
    struct.pack_into(e + "h", hdr, 68, intent_code)
    struct.pack_into(e + "hh", hdr, 70, code, vol.dtype.itemsize * 8)
    struct.pack_into(e + "8f", hdr, 76, 1.0, pixdim[0], pixdim[1], pixdim[2], pixdim[3], 1, 1, 1)
    struct.pack_into(e + "f", hdr, 108, 352.0)
    struct.pack_into(e + "ff", hdr, 112, scl[0], scl[1])
    hdr[123] = 2 | 8
    struct.pack_into(e + "hh", hdr, 252, 0, 1)
    struct.pack_into(e + "4f", hdr, 280, pixdim[0], 0, 0, 0)
    struct.pack_into(e + "4f", hdr, 296, 0, pixdim[1], 0, 0)
    struct.pack_into(e + "4f", hdr, 312, 0, 0, pixdim[2], 0)
    hdr[344:348] = b"n+1\0"
    data = np.ascontiguousarray(vol.transpose(3, 2, 1, 0)).astype(vol.dtype.newbyteorder(e))
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")
        f.write(data.tobytes())


def read(path):
    """-> (array [x, y, z, t] float64 with scaling applied, header dict)"""
    with _open(path, "rb") as f:
        raw = f.read()
    e = "<" if struct.unpack_from("<i", raw, 0)[0] == 348 else ">"
    assert struct.unpack_from(e + "i", raw, 0)[0] == 348
    dim = struct.unpack_from(e + "8h", raw, 40)
    code, bitpix = struct.unpack_from(e + "hh", raw, 70)
    vox_offset = int(struct.unpack_from(e + "f", raw, 108)[0])
    slope, inter = struct.unpack_from(e + "ff", raw, 112)
    nx, ny, nz = dim[1], max(dim[2], 1), max(dim[3], 1)
    nt = int(np.prod([max(d, 1) for d in dim[4:1 + dim[0]]])) if dim[0] >= 4 else 1
    dt = np.dtype(DTYPES[code]).newbyteorder(e)
    arr = np.frombuffer(raw, dtype=dt, count=nx * ny * nz * nt, offset=vox_offset).reshape(nt, nz, ny, nx).transpose(3, 2, 1, 0)
    arr = arr.astype(np.float64)
    if slope != 0:
        arr = arr * slope + inter
    hdr = dict(dim=dim, datatype=code, bitpix=bitpix, pixdim=struct.unpack_from(e + "8f", raw, 76),
               intent_code=struct.unpack_from(e + "h", raw, 68)[0], magic=raw[344:348], vox_offset=vox_offset,
               sform_code=struct.unpack_from(e + "h", raw, 254)[0], srow_x=struct.unpack_from(e + "4f", raw, 280),
               cal_max=struct.unpack_from(e + "f", raw, 124)[0], cal_min=struct.unpack_from(e + "f", raw, 128)[0])
    return arr, hdr
