"""ctypes mirror of the structs in include/ppm.h (field order and types must match exactly)."""
import ctypes as C

NCOL = 32
STATS_COLS = 7
K_NAMES = ("prep", "bank", "global", "topk", "local", "insert", "final", "extract", "norms")


class RefineCfg(C.Structure):
    _fields_ = [
        ("box", C.c_int), ("pixel_size", C.c_float), ("molecular_mass_kda", C.c_float),
        ("mask_radius", C.c_float), ("res_low", C.c_float), ("res_high", C.c_float),
        ("res_signed_cc", C.c_float), ("search_mask_radius", C.c_float), ("res_search", C.c_float),
        ("angular_step", C.c_float), ("top_hits", C.c_int), ("search_range_x", C.c_float),
        ("search_range_y", C.c_float), ("global_search", C.c_int), ("local_refine", C.c_int),
        ("refine_psi", C.c_int), ("refine_theta", C.c_int), ("refine_phi", C.c_int),
        ("refine_x", C.c_int), ("refine_y", C.c_int), ("normalize", C.c_int), ("invert", C.c_int),
        ("mask_falloff", C.c_float), ("iters_hit", C.c_int), ("iters_final", C.c_int),
        ("local_angle_step", C.c_float), ("local_shift_step", C.c_float), ("symmetry", C.c_char * 8), ("band_factor", C.c_float),
        ("refine_defocus", C.c_int), ("defocus_range", C.c_float), ("defocus_step", C.c_float),
        ("focus", C.c_float * 4),
        ("use_priors", C.c_int), ("prior_mean", C.c_float * 5), ("prior_var", C.c_float * 5),
        ("res_classification", C.c_float),
    ]

    @classmethod
    def make(cls, **kw):
        focus = kw.pop("focus", None)
        priors = kw.pop("priors", None)          # (mean[5], var[5]) of psi, theta, phi (degrees), x, y (Angstrom)
        d = dict(molecular_mass_kda=0.0, res_low=0.0, res_signed_cc=0.0, search_mask_radius=0.0, res_search=0.0,
                 angular_step=15.0, top_hits=20, search_range_x=0.0, search_range_y=0.0, global_search=1,
                 local_refine=1, refine_psi=1, refine_theta=1, refine_phi=1, refine_x=1, refine_y=1, normalize=1,
                 invert=0, mask_falloff=0.0, iters_hit=0, iters_final=0, local_angle_step=0.0, local_shift_step=0.0,
                 band_factor=0.0, symmetry=b"C1", refine_defocus=0, defocus_range=500.0, defocus_step=50.0, res_classification=0.0)
        d.update(kw)
        for req in ("box", "pixel_size", "mask_radius", "res_high"):
            if req not in d:
                raise ValueError(f"ERROR: RefineCfg needs {req}")
        if not d["res_search"]:
            d["res_search"] = d["res_high"]
        if isinstance(d["symmetry"], str):
            d["symmetry"] = d["symmetry"].encode()
        c = cls(**d)
        if focus is not None:
            c.focus[:] = [float(v) for v in focus]
        if priors is not None:
            c.use_priors = 1
            c.prior_mean[:] = [float(v) for v in priors[0]]
            c.prior_var[:] = [float(v) for v in priors[1]]
        return c


class ReconCfg(C.Structure):
    _fields_ = [
        ("box", C.c_int), ("pixel_size", C.c_float), ("res_limit", C.c_float),
        ("score_weight_bfactor", C.c_float), ("score_average", C.c_float), ("score_threshold", C.c_float),
        ("normalize", C.c_int), ("invert", C.c_int), ("split_by_pind", C.c_int), ("mask_radius", C.c_float),
        ("dose_weights", C.c_void_p), ("n_dose_weights", C.c_int), ("dose_exponent", C.c_float), ("dose_transition", C.c_float),
    ]

    def set_dose_weights(self, q, exponent, transition=1.0):
        """q[t] in (0, 1] per exposure (TIND); the array is kept alive on the struct."""
        import numpy as np
        self._dose = np.ascontiguousarray(q, dtype=np.float32)
        self.dose_weights = self._dose.ctypes.data_as(C.c_void_p)
        self.n_dose_weights = int(self._dose.size)
        self.dose_exponent = float(exponent)
        self.dose_transition = float(transition)
        return self


NPCOL, NTCOL = 12, 6
CSP_PARTICLES, CSP_MICROGRAPHS = 1, 2


class CspCfg(C.Structure):
    """ppm_csp_cfg (include/ppm.h)."""
    _fields_ = [("unit", C.c_int), ("refine_rotation", C.c_int), ("refine_translation", C.c_int), ("tol_angle", C.c_float * 3),
                ("tol_shift", C.c_float), ("step_tolerance", C.c_float), ("max_iterations", C.c_int), ("tind_min", C.c_int),
                ("tind_max", C.c_int), ("first", C.c_int), ("last", C.c_int), ("refine_defocus", C.c_int),
                ("defocus_range", C.c_float), ("defocus_step", C.c_float)]

    @classmethod
    def make(cls, unit, refine_rotation=1, refine_translation=1, tol_angle=(30.0, 30.0, 30.0), tol_shift=20.0, step_tolerance=0.01,
             max_iterations=0, tind_min=0, tind_max=-1, first=0, last=-1, refine_defocus=0, defocus_range=750.0, defocus_step=50.0):
        return cls(unit=int(unit), refine_rotation=int(refine_rotation), refine_translation=int(refine_translation),
                   tol_angle=(C.c_float * 3)(*[float(x) for x in tol_angle]), tol_shift=float(tol_shift),
                   step_tolerance=float(step_tolerance), max_iterations=int(max_iterations), tind_min=int(tind_min),
                   tind_max=int(tind_max), first=int(first), last=int(last), refine_defocus=int(refine_defocus),
                   defocus_range=float(defocus_range), defocus_step=float(defocus_step))


class SvaCfg(C.Structure):
    """ppm_sva_cfg (include/ppm.h)."""
    _fields_ = [("box", C.c_int), ("pixel_size", C.c_float), ("window", C.c_float * 3), ("window_sigma", C.c_float),
                ("highpass_cutoff", C.c_float), ("highpass_decay", C.c_float), ("lowpass_cutoff", C.c_float), ("lowpass_decay", C.c_float),
                ("use_missing_wedge", C.c_int), ("tol_angle", C.c_float), ("tol_shift", C.c_float), ("step_tolerance", C.c_float),
                ("max_iterations", C.c_int), ("band_factor", C.c_float), ("search_mode", C.c_int), ("global_step", C.c_float), ("n_candidates", C.c_int)]

    @classmethod
    def make(cls, box, pixel_size=1.0, window=(0, 0, 0), window_sigma=0.0, highpass=(0.0, 0.0), lowpass=(0.25, 0.05), use_missing_wedge=1,
             tol_angle=10.0, tol_shift=5.0, step_tolerance=0.05, max_iterations=0, band_factor=0.0, search_mode=0, global_step=0.0, n_candidates=0):
        return cls(box=int(box), pixel_size=float(pixel_size), window=(C.c_float * 3)(*[float(x) for x in window]), window_sigma=float(window_sigma),
                   highpass_cutoff=float(highpass[0]), highpass_decay=float(highpass[1]), lowpass_cutoff=float(lowpass[0]),
                   lowpass_decay=float(lowpass[1]), use_missing_wedge=int(use_missing_wedge), tol_angle=float(tol_angle),
                   tol_shift=float(tol_shift), step_tolerance=float(step_tolerance), max_iterations=int(max_iterations), band_factor=float(band_factor),
                   search_mode=int(search_mode), global_step=float(global_step), n_candidates=int(n_candidates))


class FinalCfg(C.Structure):
    _fields_ = [("molecular_mass_kda", C.c_float), ("inner_radius", C.c_float), ("outer_radius", C.c_float),
                ("mask_falloff", C.c_float)]
