"""Minimal PNG writer/reader (8-bit RGB / grey).  The reference writes every rendered view with imageio.imwrite
(RN:250, RN:206) and the detector reads them back (NM:667-680); imageio is not a dependency of this package, so
the same files are produced with zlib + struct.  If imageio is importable it is used instead."""
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def imwrite(path, img):
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8:
        raise TypeError("imwrite expects uint8 (use to8b)")
    try:
        import imageio
        imageio.imwrite(path, img)
        return
    except ImportError:
        pass
    if img.ndim == 2:
        h, w = img.shape
        ctype, ch = 0, 1
    elif img.ndim == 3 and img.shape[2] in (3, 4):
        h, w, ch = img.shape
        ctype = 2 if ch == 3 else 6
    else:
        raise ValueError("imwrite: unsupported shape %s" % (img.shape,))
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * ch)], 1).tobytes()   # filter 0 per row
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        f.write(_chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(_chunk(b"IEND", b""))


def imread(path):
    """Reads back what imwrite produced (filter type 0 only) -- used by the tests."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert depth == 8
            ch = {0: 1, 2: 3, 6: 4}[ctype]
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    assert (rows[:, 0] == 0).all(), "only filter type 0 is supported"
    img = rows[:, 1:].reshape(h, w, ch)
    return img[..., 0] if ch == 1 else img
