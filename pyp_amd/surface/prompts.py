"""Positional stdin answer scripts of the drop-in executables.

The caller never parses prompts; it pipes a here-doc of answers, one per line, `yes`/`no` for
booleans (src/pyp/refine/frealign/frealign.py:3918-3994 refine3d, :1780-1824 reconstruct3d,
:1878-1888 local_merge3d, :2075-2093 merge3d; the `.par`-surface refine3d script is preserved at
src/pyp/system/wrapper_functions.py:512-561).  Answer order: SURVEY.md §9.1-§9.3.
"""


class PromptError(ValueError):
    pass


def read_answers(stream):
    """All lines of the here-doc, stripped; stops at a line reading 'eot' if the shell left it in."""
    out = []
    for line in stream.read().splitlines():
        s = line.strip()
        if s == "eot":
            break
        out.append(s)
    while out and out[-1] == "":
        out.pop()
    return out


def split_heredoc(command):
    """A command string exactly as PYP builds it — '<dir>/<program> << eot >> <logfile> 2>&1\n<answer>\n...eot\n'
    (frealign.py:3918-3994, :1780-1824, :1878-1888, :2075-2093) — taken apart into (program, logfile, answers)."""
    head, _, rest = command.lstrip("\n").partition("\n")
    prog, sep, tail = head.partition("<<")
    if not sep:
        raise PromptError(f"ERROR: not a here-doc command: '{head}'")
    log = tail.split(">>", 1)[1].replace("2>&1", "").strip() if ">>" in tail else ""
    import io
    return prog.strip(), log, read_answers(io.StringIO(rest))


def _bool(s, what):
    t = s.strip().lower()
    if t in ("yes", "y", "true", "1"):
        return True
    if t in ("no", "n", "false", "0"):
        return False
    raise PromptError(f"ERROR: answer for '{what}' must be yes or no, got '{s}'")


def _num(s, what, typ=float):
    try:
        return typ(float(s)) if typ is int else typ(s)
    except ValueError:
        raise PromptError(f"ERROR: answer for '{what}' must be a number, got '{s}'")


REFINE3D_CISTEM = [  # (key, kind) in script order, 50 answers
    ("stack", str), ("input_params", str), ("global_stats", str), ("reference", str), ("statistics", str),
    ("use_statistics", bool), ("use_priors", bool), ("match_out", str), ("output_params", str), ("output_changes", str),
    ("symmetry", str), ("first", int), ("last", int), ("fraction", float), ("pixel_size", float), ("molecular_mass", float),
    ("inner_radius", float), ("outer_radius", float), ("res_low", float), ("res_high", float), ("res_signed_cc", float),
    ("res_classification", float), ("search_mask_radius", float), ("res_search", float), ("angular_step", float),
    ("top_hits", int), ("search_range_x", float), ("search_range_y", float), ("focus_x", float), ("focus_y", float),
    ("focus_z", float), ("focus_r", float), ("defocus_range", float), ("defocus_step", float), ("padding", float),
    ("global_search", bool), ("local_refine", bool), ("refine_psi", bool), ("refine_theta", bool), ("refine_phi", bool),
    ("refine_x", bool), ("refine_y", bool), ("calc_match", bool), ("mask_2d", bool), ("refine_defocus", bool),
    ("normalize", bool), ("invert", bool), ("exclude_edges", bool), ("normalize_reference", bool), ("threshold_reference", bool),
]

REFINE3D_PAR = [  # frealign_v9.11 surface, 45 answers
    ("stack", str), ("input_params", str), ("reference", str), ("statistics", str), ("use_statistics", bool),
    ("match_out", str), ("output_params", str), ("output_changes", str), ("symmetry", str), ("first", int), ("last", int),
    ("pixel_size", float), ("voltage", float), ("cs", float), ("amplitude_contrast", float), ("molecular_mass", float),
    ("outer_radius", float), ("res_low", float), ("res_high", float), ("res_signed_cc", float), ("res_classification", float),
    ("search_mask_radius", float), ("res_search", float), ("angular_step", float), ("top_hits", int),
    ("search_range_x", float), ("search_range_y", float), ("focus_x", float), ("focus_y", float), ("focus_z", float),
    ("focus_r", float), ("defocus_range", float), ("defocus_step", float), ("padding", float), ("global_search", bool),
    ("local_refine", bool), ("refine_psi", bool), ("refine_theta", bool), ("refine_phi", bool), ("refine_x", bool),
    ("refine_y", bool), ("calc_match", bool), ("mask_2d", bool), ("refine_defocus", bool), ("invert", bool),
]

MERGE3D = [("half1", str), ("half2", str), ("filtered", str), ("statistics", str), ("molecular_mass", float),
           ("inner_radius", float), ("outer_radius", float), ("dump_seed_1", str), ("dump_seed_2", str), ("n_dumps", int)]

LOCAL_MERGE3D = [("out_dump_1", str), ("out_dump_2", str), ("dump_seed_1", str), ("dump_seed_2", str), ("n_dumps", int)]


def _take(answers, spec, prog):
    if len(answers) < len(spec):
        raise PromptError(f"ERROR: {prog}: expected {len(spec)} answers, got {len(answers)}")
    out = {}
    for (key, kind), s in zip(spec, answers):
        if kind is bool:
            out[key] = _bool(s, key)
        elif kind is str:
            if s == "":
                raise PromptError(f"ERROR: {prog}: empty answer for '{key}'")
            out[key] = s
        else:
            out[key] = _num(s, key, kind)
    return out


def parse_refine3d(answers):
    """Selects the .par surface when the input parameter file is not a .cistem file."""
    if len(answers) >= 2 and not answers[1].endswith(".cistem"):
        d = _take(answers, REFINE3D_PAR, "refine3d")
        d["surface"] = "par"
        d.update(global_stats="null", use_priors=False, fraction=1.0, inner_radius=0.0, normalize=True,
                 exclude_edges=False, normalize_reference=False, threshold_reference=False)
    else:
        d = _take(answers, REFINE3D_CISTEM, "refine3d")
        d["surface"] = "cistem"
    if d["first"] < 1 or d["last"] < d["first"]:
        raise PromptError(f"ERROR: refine3d: bad particle range {d['first']}..{d['last']}")
    return d


def parse_reconstruct3d(answers):
    """39 answers, or 43 when dose weighting is switched on (its answer expands to five lines,
    frealign.py:1731-1753)."""
    a = list(answers)
    head = [("stack", str), ("input_params", str), ("global_stats", str), ("reference", str), ("map1", str), ("map2", str),
            ("output", str), ("res_file", str), ("symmetry", str), ("first", int), ("last", int), ("pixel_size", float),
            ("molecular_mass", float), ("inner_radius", float), ("outer_radius", float), ("res_limit", float),
            ("res_reference", float), ("score_bfactor", float), ("score_weighting", bool), ("min_tilt_score", float),
            ("max_tilt_score", float), ("dose_weighting", bool)]
    tail = [("score_threshold", float), ("smoothing", float), ("padding", float), ("normalize", bool), ("adjust_scores", bool),
            ("invert", bool), ("exclude_edges", bool), ("crop", bool), ("split_even_odd", bool), ("per_particle_splitting", bool),
            ("center_mass", bool), ("likelihood_blurring", bool), ("threshold_reference", bool), ("dump", bool),
            ("dump_1", str), ("dump_2", str), ("threads", int)]
    d = _take(a, head, "reconstruct3d")
    rest = a[len(head):]
    if d["dose_weighting"]:
        dw = _take(rest, [("dose_weights_file", str), ("dose_multiply", bool), ("dose_fraction", float), ("dose_transition", float)],
                   "reconstruct3d")
        d.update(dw)
        rest = rest[4:]
    d.update(_take(rest, tail, "reconstruct3d"))
    if d["first"] < 1 or d["last"] < d["first"]:
        raise PromptError(f"ERROR: reconstruct3d: bad particle range {d['first']}..{d['last']}")
    return d


def parse_merge3d(answers):
    return _take(answers, MERGE3D, "merge3d")


def parse_local_merge3d(answers):
    return _take(answers, LOCAL_MERGE3D, "local_merge3d")


def dump_name(seed, k):
    """'…_map1_n.mrc', 3 -> '…_map1_n3.mrc' (the index goes before '.mrc', frealign.py:1870-1876)."""
    if seed.endswith(".mrc"):
        return seed[:-4] + str(k) + ".mrc"
    return seed + str(k)
