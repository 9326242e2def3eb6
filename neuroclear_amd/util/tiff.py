"""Minimal TIFF reader / writer for image volumes (grayscale uint8 / uint16 / float32, 2-D pages stacked into 3-D).

The reference saves and loads volumes through scikit-image's tifffile plugin (`skimage.io.imsave / imread`,
test_dice.py:150-158,234; data/image_folder.py) -- neither scikit-image nor tifffile exists on this image, and a volume
file is just a container, so the container is written here directly from the TIFF 6.0 / BigTIFF specifications:
baseline, uncompressed, one strip per page, little-endian; BigTIFF (64-bit offsets) when the file would pass 4 GiB
(a 900^3 uint16 volume is 1.46 GB: classic).  Files written here open in ImageJ / tifffile / PIL; `imread` reads what
`imsave` writes plus any uncompressed strip-based grayscale TIFF (either byte order, classic or BigTIFF)."""
import struct

import numpy as np

_TYPES = {1: 'B', 2: 'c', 3: 'H', 4: 'I', 5: 'II', 16: 'Q'}
_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}
_SAMPLE_FORMAT = {'u': 1, 'i': 2, 'f': 3}


def imsave(path, arr):
    """arr: [H, W] or [pages, H, W]; uint8, uint16 or float32."""
    a = np.ascontiguousarray(arr)
    if a.ndim == 2:
        a = a[None]
    if a.ndim != 3:
        raise ValueError('imsave: 2-D image or 3-D stack expected, got shape %s' % (arr.shape,))
    if a.dtype not in (np.uint8, np.uint16, np.float32):
        raise TypeError('imsave: uint8 / uint16 / float32 only, got %s' % a.dtype)
    a = a.astype(a.dtype.newbyteorder('<'), copy=False)
    pages, h, w = a.shape
    page_bytes = h * w * a.dtype.itemsize
    big = pages * (page_bytes + 256) + 64 >= (1 << 32)
    ntags = 10
    if big:
        head = struct.pack('<2sHHHQ', b'II', 43, 8, 0, 16)
        ifd_size = 8 + ntags * 20 + 8
    else:
        head = struct.pack('<2sHI', b'II', 42, 8)
        ifd_size = 2 + ntags * 12 + 4
    hlen = len(head)
    data0 = hlen + pages * ifd_size  # all IFDs first, then the pixel data: one seek-free write each
    data0 = (data0 + 15) & ~15
    with open(path, 'wb') as f:
        f.write(head)
        for p in range(pages):
            off = data0 + p * page_bytes
            nxt = hlen + (p + 1) * ifd_size if p + 1 < pages else 0
            tags = [(256, 4, w), (257, 4, h), (258, 3, 8 * a.dtype.itemsize), (259, 3, 1), (262, 3, 1),
                    (273, 16 if big else 4, off), (277, 3, 1), (278, 4, h), (279, 16 if big else 4, page_bytes),
                    (339, 3, _SAMPLE_FORMAT[a.dtype.kind])]
            if big:
                f.write(struct.pack('<Q', ntags))
                for tag, typ, val in tags:
                    f.write(struct.pack('<HHQ', tag, typ, 1) + struct.pack('<Q', val))
                f.write(struct.pack('<Q', nxt))
            else:
                f.write(struct.pack('<H', ntags))
                for tag, typ, val in tags:
                    f.write(struct.pack('<HHI', tag, typ, 1) + (struct.pack('<HH', val, 0) if typ == 3 else struct.pack('<I', val)))
                f.write(struct.pack('<I', nxt))
        f.write(b'\0' * (data0 - f.tell()))
        a.tofile(f)


def imread(path):
    """-> [H, W] (one page) or [pages, H, W]."""
    with open(path, 'rb') as f:
        buf = f.read(16)
        bo = {b'II': '<', b'MM': '>'}.get(buf[:2])
        if bo is None:
            raise ValueError('%s: not a TIFF file' % path)
        magic = struct.unpack(bo + 'H', buf[2:4])[0]
        if magic == 42:
            big, off = False, struct.unpack(bo + 'I', buf[4:8])[0]
        elif magic == 43:
            big, off = True, struct.unpack(bo + 'Q', buf[8:16])[0]
        else:
            raise ValueError('%s: bad TIFF magic %d' % (path, magic))
        pages = []
        while off:
            f.seek(off)
            n = struct.unpack(bo + ('Q' if big else 'H'), f.read(8 if big else 2))[0]
            esz = 20 if big else 12
            raw = f.read(n * esz)
            nxt = struct.unpack(bo + ('Q' if big else 'I'), f.read(8 if big else 4))[0]
            tags = {}
            for i in range(n):
                e = raw[i * esz:(i + 1) * esz]
                tag, typ = struct.unpack(bo + 'HH', e[:4])
                cnt = struct.unpack(bo + ('Q' if big else 'I'), e[4:12] if big else e[4:8])[0]
                vfield = e[12:20] if big else e[8:12]
                size = _SIZES.get(typ, 1) * cnt
                if size <= len(vfield):
                    data = vfield[:size]
                else:
                    pos = f.tell()
                    f.seek(struct.unpack(bo + ('Q' if big else 'I'), vfield)[0])
                    data = f.read(size)
                    f.seek(pos)
                code = {1: 'B', 3: 'H', 4: 'I', 16: 'Q'}.get(typ)
                tags[tag] = list(struct.unpack(bo + code * cnt, data)) if code else data
            if tags.get(259, [1])[0] != 1:
                raise NotImplementedError('%s: compressed TIFF (compression %d) is not supported' % (path, tags[259][0]))
            if tags.get(277, [1])[0] != 1:
                raise NotImplementedError('%s: only single-sample (grayscale) pages are supported' % path)
            w, h, bits = tags[256][0], tags[257][0], tags.get(258, [8])[0]
            kind = {1: 'u', 2: 'i', 3: 'f'}[tags.get(339, [1])[0]]
            dt = np.dtype('%s%s%d' % (bo, kind, bits // 8))
            page = np.empty(h * w, dtype=dt)
            filled = 0
            for so, sb in zip(tags[273], tags[279]):
                f.seek(so)
                cnt_ = sb // dt.itemsize
                page[filled:filled + cnt_] = np.frombuffer(f.read(sb), dtype=dt, count=cnt_)
                filled += cnt_
            if filled != h * w:
                raise ValueError('%s: strips hold %d of %d pixels' % (path, filled, h * w))
            pages.append(page.reshape(h, w).astype(dt.newbyteorder('='), copy=False))
            off = nxt
    if not pages:
        raise ValueError('%s: no image' % path)
    return pages[0] if len(pages) == 1 else np.stack(pages)
