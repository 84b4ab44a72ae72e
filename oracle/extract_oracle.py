"""numpy restatement of PYP's particle extraction + box normalisation (TEST INFRASTRUCTURE ONLY).

Follows src/pyp/extract/core.py:447-506 (window bounds, fill with the inside mean) and
src/pyp/analysis/image.py:320-340 (extract_background), :406-417 (normalize_image), :461-471
(fix_empty_particles_in_place).  The normalisation part is pinned by tests/golden/normalize_image.npz, which was
produced by running the reference's own normalize_image / extract_background.  The reference replaces empty boxes
with UNSEEDED numpy noise, which cannot be pinned; `is_empty` returns the same decision instead.
"""
import math

import numpy as np


def window(image, bx, by, boxsize, coordinate_binning):
    """raw box (float64) around (box[0] = bx -> columns, box[1] = by -> rows), extract/core.py:447-491."""
    nx, ny = image.shape[-2], image.shape[-1]
    minx = miny = 0
    maxx = maxy = boxsize
    minX = math.floor(by / float(coordinate_binning) - math.floor(boxsize / 2.0))
    maxX = minX + boxsize
    minY = math.floor(bx / float(coordinate_binning) - math.floor(boxsize / 2.0))
    maxY = minY + boxsize
    if minX < 0:
        minx = -minX
        minX = 0
    elif maxX >= nx:
        maxx = -(maxX - nx + 1)
        maxX = nx - 1
    if minY < 0:
        miny = -minY
        minY = 0
    elif maxY >= ny:
        maxy = -(maxY - ny + 1)
        maxY = ny - 1
    inside = np.squeeze(image[int(minX):int(maxX), int(minY):int(maxY)])
    if inside.ndim == 2 and min(inside.shape) > 0:
        raw = inside.mean() * np.ones([boxsize, boxsize])
        raw[int(minx):int(maxx), int(miny):int(maxy)] = inside
    else:
        raw = np.zeros([boxsize, boxsize])
    return raw


def is_empty(frame):
    return bool(frame.min() == frame.max()
                or np.where(frame == np.median(frame), 0, 1).sum() < frame.shape[0] * frame.shape[1] * 0.01)


def background(image, radius, pixelsize):
    boxsize = image.shape[1]
    x, y = np.mgrid[0:boxsize, 0:boxsize] - boxsize // 2
    if radius / pixelsize > boxsize / 2:
        radius = boxsize * pixelsize / 2
    cond = np.hypot(x, y) > radius / pixelsize
    bg = np.extract(cond, image)
    return [bg.mean(), bg.std()]


def normalize_image(image, radius, pixelsize, binning):
    mean, std = background(image, radius, pixelsize * binning)
    out = image - mean
    if std > 0:
        out /= std
    return out


def extract(image, coords, boxsize, radius, pixelsize, coordinate_binning=1, normalize=True):
    """(M, box, box) float32 stack and the per-box emptiness decisions."""
    out = np.empty((len(coords), boxsize, boxsize), dtype=np.float32)
    empty = []
    for i, (bx, by) in enumerate(coords):
        raw = window(image, bx, by, boxsize, coordinate_binning)
        empty.append(is_empty(raw))
        if normalize and not empty[-1]:
            raw = normalize_image(raw, radius, pixelsize, coordinate_binning)
        out[i] = raw
    return out, empty
