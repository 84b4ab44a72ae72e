"""(Test infrastructure: the CALLER's side of region-based refinement, restated so that tests can drive bin/csp the way PYP does; nothing in pyp_amd imports it.)
Region-based ("patch") constrained refinement, the caller's side: a tilt series is cut into a
grid of regions, every region gets its own parameter file whose tilts are refined on the region's particles only, and `csp`
is started once per (region, particle) or (region, tilt).  numpy restatement of

    src/pyp/analysis/geometry/core.py:632-680   findSpecimenBounds
    src/pyp/analysis/geometry/core.py:554-630   divide2regions
    src/pyp/refine/csp/particle_cspt.py:34-93   sort_particles_regions
    src/pyp/refine/csp/particle_cspt.py:141-208 split_parameter_file
    src/pyp/system/local_run.py:306-467         create_csp_split_commands (both branches)

PINNED by tests/golden/golden_r03.json ("regions", "csp_split_commands"): bounds, corners, the sorted particle lists, the
region files byte for byte and every command line are the reference's own output on a toy series (tests/test_regions.py).
Particle blocks are [P, 12] arrays in `cistem.PARTICLE_COLUMNS` order (3-D position in columns 7-9), tilt blocks [T, 6] in
`cistem.TILT_COLUMNS` order (TIND, RIND, shift x, shift y, angle, axis).
"""

import math
import os

import numpy as np

from pyp_amd.formats import cistem

C = cistem.COL


def find_specimen_bounds(particles, dim_tomogram):
    """z bounds from the particle positions (floor / ceil, the reference's `if … elif …` update included: a coordinate that
    lowers the minimum is not also tried as a maximum); x and y span the whole tomogram."""
    min_z, max_z = dim_tomogram[2], 0
    for p in np.asarray(particles, dtype=np.float64):
        z = p[9]
        if z < min_z:
            min_z = math.floor(z)
        elif z > max_z:
            max_z = math.ceil(z)
    return [0, 0, min_z], [dim_tomogram[0], dim_tomogram[1], max_z]


def divide_regions(bottom_left, top_right, split_x=4, split_y=4, split_z=1, overlap=0.0):
    """Corners (x outermost, then y, then z) and the common size of the grid's boxes."""
    if split_x < 1 or split_y < 1 or split_z < 1:
        raise ValueError("ERROR: split x/y/z has to be greater than zero")
    size, inc = [], []
    for lo, hi, n in zip(bottom_left, top_right, (split_x, split_y, split_z)):
        den = n - n * overlap + overlap
        s = (hi - lo) / den if den != 0 else (hi - lo)
        size.append(s)
        inc.append(s * (1 - overlap))
    corners = [[bottom_left[0] + inc[0] * i, bottom_left[1] + inc[1] * j, bottom_left[2] + inc[2] * k]
               for i in range(split_x) for j in range(split_y) for k in range(split_z)]
    return corners, size


def sort_particles_regions(particles, corners, size, per_particle=False):
    """Lists of particle indices per region, plus one trailing list for particles outside every box, sorted by length
    (stable, so equal lengths keep the grid order); a particle goes to the FIRST box that contains it (closed intervals)."""
    particles = np.asarray(particles, dtype=np.float64)
    if per_particle:
        return sorted([[int(p[0])] for p in particles], key=len)
    ret = [[] for _ in range(len(corners) + 1)]
    for p in particles:
        x, y, z = p[7], p[8], p[9]
        for k, c in enumerate(corners):
            if c[0] <= x <= c[0] + size[0] and c[1] <= y <= c[1] + size[1] and c[2] <= z <= c[2] + size[2]:
                ret[k].append(int(p[0]))
                break
        else:
            ret[-1].append(int(p[0]))
    return sorted(ret, key=len)


def split_parameter_file(rows, particles, tilts, parameter_file, regions):
    """Write `<parameter_file>_regionNNNN.cistem` (+ `_extended`) for every non-empty region, numbered in list order, and return
    [(file, PINDs, TINDs)].  A region file: the rows of its particles with RIND = the new region index; extended: ALL
    particles; the tilt entries (TIND, old RIND) its rows use, re-keyed (TIND, new index)."""
    rows = np.asarray(rows, dtype=np.float64)
    tilts = np.asarray(tilts, dtype=np.float64)
    tmap = {(int(t[0]), int(t[1])): t for t in tilts}
    out, k = [], 0
    for region in regions:
        if len(region) == 0:
            continue
        sub = rows[np.isin(rows[:, C["PIND"]], region)]
        if sub.size == 0:
            continue
        pairs = np.unique(sub[:, [C["TIND"], C["RIND"]]].astype(np.int64), axis=0)
        tb = []
        for tind, rind in pairs:
            if (int(tind), int(rind)) not in tmap:
                raise ValueError(f"ERROR: tilt (TIND {tind}, RIND {rind}) is missing from the extended parameters")
            t = tmap[(int(tind), int(rind))].copy()
            t[1] = k
            tb.append(t)
        sub = sub.copy()
        sub[:, C["RIND"]] = k
        fn = parameter_file.replace(".cistem", "_region%04d.cistem" % k)
        cistem.write_parameters(fn, sub)
        cistem.write_extended(fn.replace(".cistem", "_extended.cistem"), particles, np.array(tb))
        out.append((fn, np.unique(sub[:, C["PIND"]].astype(np.int64)), np.unique(sub[:, C["TIND"]].astype(np.int64))))
        k += 1
    return out


def csp_split_commands(csp_command, parameter_file, mode, name, merged_stack, ptlind_list, scanord_list, increment=1, use_frames=False,
                       frame_refinement=False):
    """The command lines of create_csp_split_commands.  `parameter_file`: a path (global branch; `increment` = particles per
    job, which the reference derives from the memory budget, local_run.py:415-423) or the list split_parameter_file
    returned (region branch).  Returns (commands, movie_list)."""
    name = name.split("_r")[0]
    images = "frames_csp.txt" if use_frames else os.path.join("frealign", "%s.mrc" % name)
    commands, movies = [], []
    fmt = "{0} {1} {2} {3} {4} {5} {6} {7} {8} > {9}"
    if isinstance(parameter_file, list):
        refine_frames = "1" if (not use_frames or frame_refinement) else "0"
        if mode == 3 and not frame_refinement:
            mode = 6
        if mode == 2:
            mode = 5
        for core, region in enumerate(parameter_file[::-1]):
            split_file = region[0]
            tag = split_file.split("region")[-1].split("_")[0]
            if mode in (3, 6, 4):
                firsts = [0] if frame_refinement else list(region[2])
                lasts = [-1] if frame_refinement else list(region[2])
            else:
                firsts = lasts = list(region[1])
            for a, b in zip(firsts, lasts):
                log = "%s_csp_region%s_%06d_%06d.log" % (name, tag, a, b) if core == 0 else "/dev/null"
                commands.append(fmt.format(csp_command, split_file, split_file.replace(".cistem", "_extended.cistem"), mode, int(a), int(b),
                                           refine_frames, images, merged_stack, log))
        return commands, movies
    ext = parameter_file.replace(".cistem", "_extended.cistem")
    extract_frame = 1
    if mode in (2, -2):
        mode = 5 if mode == 2 else mode
        units = list(ptlind_list)
        if use_frames and mode == 5:
            increment = 1
    elif mode == 3 and not use_frames:
        mode, units, increment = 6, list(scanord_list), 1
    elif mode == 3:
        extract_frame, units, increment = 0, list(ptlind_list), 1
    else:
        raise ValueError(f"ERROR: csp_split_commands: mode {mode} has no global branch")
    for i0 in range(0, len(units), increment):
        first, last = int(units[i0]), int(units[min(i0 + increment - 1, len(units) - 1)])
        log = "%s_csp_%06d_%06d.log" % (name, first, last) if first == 0 else "/dev/null"
        stack = merged_stack if mode != -2 else "frealign/%s_stack_%04d_%04d.mrc" % (name, first, last)
        commands.append(fmt.format(csp_command, parameter_file, ext, mode, first, last, extract_frame, images, stack, log))
        movies.append("frealign/%s_stack_%04d_%04d.mrc" % (name, first, last))
    return commands, movies
