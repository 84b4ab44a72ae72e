/* ppm.h — C ABI of libpypmatch.so: per-particle projection matching and Fourier-slice
 * insertion on one MI355X (gfx950).  This is the drop-in boundary for the numerics PYP
 * shells out to external binaries (none of which ship as source):
 *
 *   ppm_refine_batch      replaces the per-range `refine3d` process that
 *                         src/pyp/refine/frealign/frealign.py:3918-3994 (mrefine_version)
 *                         scripts and src/pyp/system/local_run.py:472-579 fans out; the
 *                         `.par` twin is src/pyp/system/wrapper_functions.py:512-561.
 *   ppm_insert_batch      replaces the per-range `reconstruct3d` process scripted at
 *                         src/pyp/refine/frealign/frealign.py:1780-1824 (split_reconstruction).
 *   ppm_accum_add /       replace `local_merge3d` (frealign.py:1878-1888): sum of dump pairs.
 *   ppm_accum_download
 *   ppm_finalize          replaces `merge3d` (frealign.py:2075-2093): FSC / part-FSC / SSNR table,
 *                         Wiener-filtered map and the two half maps.
*   ppm_sva_align /       replace the alignment and averaging steps of `external/TOMO/MPI_Classification`
 *   ppm_sva_insert        (src/pyp/refine/tomo_avg/sub_tomo_avg.py:468-555, src/pyp_main.py:3007-3107).
 *   ppm_reference_create  is the "input reconstruction" preparation both binaries do at start-up
 *                         (answer 4 of the refine3d script, frealign.py:3923).
 *
 * A parameter row is the 32-column `.cistem` row in file order
 * (src/pyp/inout/metadata/cistem_star_file.py:596-628), held as doubles like the reference
 * holds it in RAM (:716-727).  Angles in degrees, shifts in Angstrom (src/pyp/analysis/scores.py:693).
 *
 * Conventions: all functions return 0 on success, a negative errno-style code on failure and
 * leave a message for ppm_last_error() (thread-local).  The caller owns every host buffer; the library owns
 * device memory.  No torch types cross this boundary.
 *
 * Process model: one process per GPU.  ppm_init binds the process to one device (a second call with another index
 * fails).  Thread-compatible per handle: every reference / accumulator handle owns its HIP streams and workspaces, so calls on
 * DIFFERENT handles may be made concurrently from different threads (two class references refined side by side, an insertion
 * running next to a refinement); calls on ONE handle must be serialised by the caller.  Process-wide tables (FFT plans, the
 * profiling counters) are guarded inside the library.  Entry points without a handle (ppm_extract_boxes, ppm_device_*,
 * ppm_host_*) are thread-safe; ppm_extract_boxes calls share one stream and are serialised on the device.
 *
 * Stream ordering: device buffers handed in (resident particle stacks, an external accumulator buffer, extraction
 * outputs) are read / written on the handle's stream, which is NOT ordered against any stream of the caller: finish
 * (or synchronise) the work that produces them before the call; every entry point returns only after its own device
 * work has completed, so results may be used on any stream afterwards.
 */
#ifndef PPM_H
#define PPM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPM_NCOL 32 /* columns of a .cistem row */
/* column indices inside a row (cistem_star_file.py:596-628) */
enum {
    PPM_POS = 0, PPM_PSI = 1, PPM_THETA = 2, PPM_PHI = 3, PPM_XSHIFT = 4, PPM_YSHIFT = 5,
    PPM_DF1 = 6, PPM_DF2 = 7, PPM_ANGAST = 8, PPM_PSHIFT = 9, PPM_FILM = 10, PPM_OCC = 11,
    PPM_LOGP = 12, PPM_SIGMA = 13, PPM_SCORE = 14, PPM_PIXEL = 15, PPM_VOLTAGE = 16,
    PPM_CS = 17, PPM_AMP = 18, PPM_BTX = 19, PPM_BTY = 20, PPM_PIND = 26, PPM_TIND = 27
};
/* BEAM_TILT_X / BEAM_TILT_Y (milliradians): every particle spectrum is multiplied by exp(-i phi), phi(s) = 2 pi Cs lambda^2 |s|^2 (s . b), when
 * it is prepared (k_prep), i.e. the model is compared with, and the reconstruction receives, the image with the beam-tilt phase error
 * removed.  The sign follows the usual convention (the tilted beam multiplies the image transform by exp(+i phi)); the absent programs'
 * own convention is not visible: unpinned like the rest of the arithmetic. */

#define PPM_MAX_SHIFT_STEPS 8 /* half-width of the global-search shift window, in search-grid steps */
#define PPM_MAX_TOP_HITS 64
#define PPM_MAX_DEFOCUS_STEPS 20 /* defocus offsets tried on either side when refine_defocus is set */

/* Refinement settings = the numeric answers of the refine3d prompt script
 * (frealign.py:3918-3994; answer numbers as in SURVEY.md §9.1). */
typedef struct ppm_refine_cfg {
    int box;                  /* particle box edge N (power of two, 32..512) */
    float pixel_size;         /* 15: Angstrom per pixel */
    float molecular_mass_kda; /* 16 (kept for the log; not used by matching) */
    float mask_radius;        /* 18: outer mask radius, Angstrom */
    float res_low;            /* 19: low-resolution limit, Angstrom (0 = none) */
    float res_high;           /* 20: high-resolution limit, Angstrom */
    float res_signed_cc;      /* 21: rings at lower resolution than this are summed signed, the
                                 rest by absolute value; 0 = all signed */
    float search_mask_radius; /* 23: mask radius for the global search, Angstrom (0 = mask_radius) */
    float res_search;         /* 24: resolution limit of the global search, Angstrom; the grid search itself never uses more
                                 than 64 Fourier pixels (a finer limit is lowered to that and ppm_refine_note() says so; the hits
                                 are still refined up to res_high) */
    float angular_step;       /* 25: degrees */
    int top_hits;             /* 26: global-search hits that get refined (the caller passes 20) */
    float search_range_x;     /* 27: Angstrom, 0 = widest supported window */
    float search_range_y;     /* 28 */
    int global_search;        /* 36: grid search over the asymmetric unit; its `top_hits` best orientations are ALWAYS refined
                                 (iters_hit compass iterations each, at the search band) and the best one is kept - PYP's default
                                 call is global = yes, local = no with 20 hits to refine (frealign.py:3866-3871, :3953) */
    int local_refine;         /* 37: with global_search: the best hit continues at the full band (iters_final iterations);
                                 without: refinement starts at the row's pose (iters_hit + iters_final iterations).
                                 SCORE / LOGP / SIGMA always come from one last evaluation at the full band */
    int refine_psi, refine_theta, refine_phi, refine_x, refine_y; /* 38-42 */
    int normalize;            /* 46: normalise particles */
    int invert;               /* 47: invert contrast */
    /* build-defined knobs (0 = default), DESIGN.md "search driver" */
    float mask_falloff;       /* cosine edge width of the mask, Angstrom (default 20) */
    int iters_hit;            /* compass iterations run on every hit (0 = default 2; < 0 = none: hits stay on the grid, test hook) */
    int iters_final;          /* further iterations on the best hit / on a local-only start (default 7) */
    float local_angle_step;   /* first step of a local-only refinement, degrees (default 2.5) */
    float local_shift_step;   /* same for shifts, pixels (default 2) */
    char symmetry[8];         /* 11: point group of the reference ("C1", "C7", "D7", "T", "O", "I"; "" = C1).  The global grid
                                 is restricted to phi < 360/n for Cn / Dn and theta <= 90 for Dn; T and I search the D2 unit they
                                 contain, O the D4 unit.  Every pose found is symmetry-equivalent to the unrestricted optimum */
    float band_factor;        /* frequency marching: a compass iteration whose largest probe displacement is d pixels
                                 (mask radius x angular step in radians, or the shift step) scores only rings below
                                 band_factor * N / (2 pi d), capped by the stage's band.  0 = default 3, < 0 = off */
    int refine_defocus;       /* 45: after the pose, score defocus offsets -range..+range in steps of `defocus_step` at the final
                                 pose and full band (both DEFOCUS_1 and DEFOCUS_2 move together); the best offset is added to
                                 the two columns; ties keep the smaller index (0 first) */
    float defocus_range;      /* 33: Angstrom (the caller passes 500) */
    float defocus_step;       /* 34: Angstrom (the caller passes 50); at most 20 steps either side */
    float focus[4];           /* 29-32 + 44 "apply 2D masking" (class_focusmask, frealign.py:3846-3849, :3958): centre (x, y, z) of a sphere
                                 in the reference, Angstrom FROM THE BOX CENTRE, and its radius; radius > 0 replaces the centred circular
                                 mask of every particle image by a cosine-edged disc of that radius around the sphere's projection at the
                                 row's INPUT pose: centre = (M^T c)_xy + shift (M = Rz(phi) Ry(theta) Rz(psi)).  The background
                                 statistics keep using mask_radius.  Radius <= 0: off */
    int use_priors;           /* 7 "use priors" (refine_priors, frealign.py:3841-3844, :3927) with the statistics of answer 3
                                 (`<name>_stat.cistem`: row 0 = column means, row 1 = column variances over the data set,
                                 src/pyp_main.py:2667-2674): every score the compass search compares is lowered by a Gaussian restraint
                                 sum_i (p_i - mean_i)^2 / (2 var_i n_s) over the REFINED parameters with var_i > 0, angles wrapped to
                                 +-180 degrees, n_s = pi (r_hi^2 - r_lo^2) the in-band samples of the full plane (so the restraint is a
                                 log-prior in the units of LOGP).  SCORE / LOGP / SIGMA stay those of the data term alone.  Build-defined
                                 (the absent program's rule is not visible); 0 = off */
    float prior_mean[5];      /* psi, theta, phi (degrees), x, y (Angstrom) */
    float prior_var[5];       /* their variances (degrees^2, Angstrom^2); <= 0 leaves that parameter unrestrained */
    float res_classification; /* 22 "classification resolution limit" (class_rhcls, frealign.py:3945; default 8 A,
                                 config/pyp_config.toml:5091-5095): LOGP and SIGMA of the output row - what the occupancy update of
                                 3-D classification reads (src/pyp/analysis/occupancies.py:67-252) - are evaluated at the final pose
                                 over res_low .. this limit instead of res_low .. res_high; SCORE stays that of the full band.
                                 0, or a limit beyond res_high = res_high.  ppm_csp_refine ignores it (the csp program has no such
                                 setting) */
} ppm_refine_cfg;

/* Reconstruction settings = numeric answers of the reconstruct3d script (frealign.py:1780-1824). */
typedef struct ppm_recon_cfg {
    int box;
    float pixel_size;
    float res_limit;        /* reconstruction resolution limit, Angstrom (caller passes 2*pixel) */
    float score_weight_bfactor; /* "weighting factor" refine_bsc, A^2 per score unit; 0 = off */
    float score_average;    /* mean SCORE used by the weighting (from <name>_stat.cistem or the rows) */
    float score_threshold;  /* rows with SCORE below it are skipped (caller passes 0) */
    int normalize;
    int invert;
    int split_by_pind;      /* 1: half = PIND parity, 0: half = POSITION_IN_STACK parity */
    float mask_radius;      /* outer radius for the normalisation statistics, Angstrom */
    /* data-driven dose weighting (the five-line answer of frealign.py:1731-1753): a row of exposure t = TIND is attenuated by
     * q_t ^ (dose_exponent min(1, (s / (dose_transition s_Nyquist))^2)), q_t = dose_weights[t] in (0, 1] (the exposure's mean
     * score over the best exposure's, src/pyp/inout/metadata/core.py:3039-3075); exposures beyond the table or with
     * q <= 0 are not attenuated.  NULL / 0 = off. */
    const float *dose_weights; int n_dose_weights;
    float dose_exponent;    /* "fraction" answer (x frames per exposure when "multiply" is yes); larger = fewer exposures at high resolution */
    float dose_transition;  /* fraction of Nyquist at which the full attenuation is reached (0 = 1) */
} ppm_recon_cfg;

typedef struct ppm_final_cfg {
    float molecular_mass_kda;
    float inner_radius;     /* Angstrom */
    float outer_radius;     /* Angstrom */
    float mask_falloff;     /* Angstrom (default 10) */
} ppm_final_cfg;

/* Constrained refinement of tilt-series particles (the `csp` program, src/pyp/system/local_run.py:306-467 argv,
 * mode list src/pyp/align/core.py:1015-1023): every projection row belongs to one particle (PIND) and one tilt (TIND, RIND);
 * its pose is not free but follows from the particle's 3-D pose and the tilt geometry (the relation the reference states in
 * csp_euler_angles, src/pyp/analysis/geometry/core.py:1081-1213):
 *     M_row = E(-ppsi, -ptheta, -pphi) Ry(-tilt_angle) Rz(tilt_axis),   E = Rz(phi) Ry(theta) Rz(psi)
 *     (shx, shy) = [Rz(-tilt_axis) Ry(tilt_angle) (-pshift)]_xy + (tshift_x, tshift_y)            (pixels)
 * A unit (particle or tilt) is scored by the mean score of its rows. */
enum { PPM_CSP_PARTICLES = 1, PPM_CSP_MICROGRAPHS = 2 };
#define PPM_NPCOL 12 /* particle block columns of _extended.cistem (cistem_star_file.py:247): PIND, shift x y z, psi theta phi,
                        x y z position 3-D, score, occ */
#define PPM_NTCOL 6  /* tilt block columns (:248): TIND, RIND, shift x y, angle, axis */
typedef struct ppm_csp_cfg {
    int unit;               /* PPM_CSP_PARTICLES (csp modes 1 / 2 / 5) or PPM_CSP_MICROGRAPHS (modes 0 / 3 / 6) */
    int refine_rotation;    /* particles: the three rotations (about the specimen x, y, z axes); tilts: tilt angle and tilt-axis angle */
    int refine_translation; /* particles: the 3-D shift; tilts: the two image shifts */
    float tol_angle[3];     /* search bound either side of the start, degrees (csp_ToleranceParticlesPsi / Theta / Phi ->
                               specimen x / y / z; tilts: [0] csp_ToleranceMicrographTiltAngles, [1] ...TiltAxisAngles) */
    float tol_shift;        /* the same for shifts, pixels (csp_ToleranceParticlesShifts / csp_ToleranceMicrographShifts) */
    float step_tolerance;   /* smallest compass step, degrees / pixels (csp_OptimizerStepTolerance; default 0.01) */
    int max_iterations;     /* compass iterations; 0 = until the step falls below step_tolerance, at most 12 */
    int tind_min, tind_max; /* rows with TIND outside do not enter a unit's score (csp_UseImagesForRefinementMin / Max; max < 0 = no limit) */
    int first, last;        /* units to refine: PIND (particles) or TIND (tilts) in first..last; last < 0 = up to the end */
    /* csp mode 4 (tilts only; excludes the geometric parameters): one defocus offset per tilt, -range .. +range in steps, added
     * to DEFOCUS_1 and DEFOCUS_2 of all its rows; the offset with the best mean score of the usable rows wins (the unshifted
     * values on ties, then the lower offset) */
    int refine_defocus;
    float defocus_range;    /* csp_ToleranceMicrographDefocus1, Angstrom */
    float defocus_step;     /* Angstrom (default 50); at most PPM_MAX_DEFOCUS_STEPS either side */
} ppm_csp_cfg;

#define PPM_STATS_COLS 7 /* shell, resolution A, ring radius, FSC, part-FSC, part-SSNR, rec-SSNR
                            (src/pyp/postprocess/core.py:203-221; frealign.py:2559) */

typedef struct ppm_ref ppm_ref_t;
typedef struct ppm_accum ppm_accum_t;

/* kernels whose device time the library accumulates when profiling is on */
enum { PPM_K_PREP = 0, PPM_K_BANK = 1, PPM_K_GLOBAL = 2, PPM_K_TOPK = 3, PPM_K_LOCAL = 4,
       PPM_K_INSERT = 5, PPM_K_FINAL = 6, PPM_K_EXTRACT = 7, PPM_K_NORMS = 8, PPM_K_COUNT = 9 };

int ppm_init(int device);
const char *ppm_last_error(void);
const char *ppm_version(void);
const char *ppm_build_id(void);       /* changes with every build of the library: a resident server and its clients compare it */
int ppm_device_mem_info(size_t *free_bytes, size_t *total_bytes);       /* of the library's device (hipMemGetInfo) */

/* vol: n*n*n floats, x fastest. max_band_px: largest Fourier radius (pixels) any later call will
 * use with this reference (<= n/2). */
ppm_ref_t *ppm_reference_create(const float *vol, int n, float max_band_px);
/* The same with the "padding factor" answer (refine_iblow, frealign.py:3962): the reference is zero-padded to (pad n)^3 before
 * its transform, which is then sampled pad times finer (smaller interpolation error).  pad = 1, 2 or 4, pad n <= 512. */
ppm_ref_t *ppm_reference_create_padded(const float *vol, int n, float max_band_px, int pad);
/* ... and with the "use statistics" answer (6, frealign.py:3903-3910): ring_weight[k], k = 0 .. n_weight-1, multiplies the
 * transform at |k| Fourier pixels of the unpadded box (linear interpolation; the last value beyond the table); NULL = none. */
ppm_ref_t *ppm_reference_create_weighted(const float *vol, int n, float max_band_px, int pad, const float *ring_weight, int n_weight);
void ppm_reference_destroy(ppm_ref_t *ref);

/* images: n_img * box * box floats. images_on_device != 0 means `images` is a device pointer.
 * rows_in / rows_out: n_img * PPM_NCOL doubles (host).  Image i belongs to row i. */
int ppm_refine_batch(ppm_ref_t *ref, const ppm_refine_cfg *cfg, const void *images,
                     int images_on_device, int n_img, const double *rows_in, double *rows_out);
/* per particle, for the roofline's algorithmic byte count: orientations of the global grid, local score
 * evaluations, in-band samples S(r_search) of one grid orientation, and the in-band sample counts summed
 * over all local evaluations (frequency marching makes early ones cheaper) */
int ppm_refine_last_counts(ppm_ref_t *ref, long *n_global, long *n_local, long *samples_global,
                           long *samples_local);
/* After ppm_csp_refine: 0, sweeps (k_csp_eval launches), in-band samples of the full band, gathered samples per projection summed over
 * the sweeps.  After ppm_sva_align: grid rotations of a global search, sweeps (k_sva_eval launches), samples of the band (half space,
 * before the missing wedge), band samples x gathered rotations per sub-volume summed over the sweeps. */
/* remarks of the last ppm_refine_batch on this reference that the caller should log (e.g. the search band was capped);
 * "" if none */
const char *ppm_refine_note(ppm_ref_t *ref);

/* refine3d answers 8 / 43 "matching projections" (frealign.py:3929-3931, refine_fmatch): out[i] (box * box floats, host) = the
 * reference projected at row i's pose, times the row's CTF, moved to the row's X / Y shift and band-limited at cfg->res_high — the
 * noise-free model of the stored particle image (negated when cfg->invert is set, so that it overlays the stack as stored). */
int ppm_match_projections(ppm_ref_t *ref, const ppm_refine_cfg *cfg, const double *rows, int n_rows, float *out);

/* Constrained refinement: rows (n_proj x PPM_NCOL, image i belongs to row i), particles (n_part x PPM_NPCOL) and tilts
 * (n_tilt x PPM_NTCOL) are read and updated in place: refined units get their parameters, their rows get the poses that
 * follow from them (angles from M_row, shifts moved by the change of the geometric shift) and SCORE / LOGP / SIGMA at the
 * full band; rows of units outside first..last are left untouched.  cfg gives box, pixel, mask radius, resolution limits
 * and band_factor as in ppm_refine_batch (its search fields are ignored). */
int ppm_csp_refine(ppm_ref_t *ref, const ppm_refine_cfg *cfg, const ppm_csp_cfg *csp, const void *images, int images_on_device,
                   int n_proj, double *rows, double *particles, int n_part, double *tilts, int n_tilt);

/* Sub-tomogram alignment (3DAVG, `external/TOMO/MPI_Classification`, driven by src/pyp/refine/tomo_avg/sub_tomo_avg.py:318-555
 * with the XML protocols of src/pyp/refine/3DAVG/): every sub-volume is aligned to the reference by a rotation + 3-D shift that
 * maximise the band-passed, missing-wedge-weighted normalised cross-correlation of its 3-D transform with the rotated
 * reference transform.  Pose convention = the particle block of constrained refinement: the sub-volume's transform at k
 * matches the reference's at N k, times e^{+2 pi i k.p / box} (N = E(-ppsi, -ptheta, -pphi), p = particle shift), so a refined
 * pose can be written straight into a particle's (ppsi, ptheta, pphi, shift) and vice versa. */
typedef struct ppm_sva_cfg {
    int box; float pixel_size;
    float window[3];        /* <mode>_image_window_x/y/z: half-axes of the real-space window, pixels (0 = none) */
    float window_sigma;     /* <mode>_image_window_sigma: Gaussian fall-off outside the window, pixels */
    float highpass_cutoff, highpass_decay, lowpass_cutoff, lowpass_decay;   /* <mode>_high/low_pass_cutoff/decay, cycles per pixel (Nyquist = 0.5) */
    int use_missing_wedge;  /* metric/use_missing_wedge: samples outside the tilt range lwedge..uwedge do not count */
    float tol_angle;        /* <mode>_out_of_plane_search_range: search bound either side of the start, degrees (all three rotations) */
    float tol_shift;        /* <mode>_shifts_tolerance, pixels */
    float step_tolerance;   /* smallest compass step (default 0.05 degrees / pixels) */
    int max_iterations;     /* 0 = until the step falls below step_tolerance, at most 12 */
    float band_factor;      /* frequency marching like the band_factor field of the refinement settings (0 = default 3, < 0 = off) */
    int search_mode;        /* metric/alignment_mode of the protocol (iteration_002_mode_3.xml:29-38).  0 = rotation and translation REFINEMENT
                               within the tolerances (the protocol's mode 1); 1 = GLOBAL rotation and translation search (the protocol's
                               mode 0): the start rotation times every rotation of a grid of step `global_step` over the whole of SO(3)
                               (theta_i = 180 i / (n - 1), n_phi = round(360 sin theta / step), n_psi = round(360 / step)) is ranked by the
                               correlation of the AMPLITUDES |F(k)| and |Ref(N k)| - a shift only moves phases, so the ranking does not
                               depend on how well the sub-volume is centred - on the coarse band the step allows (frequency marching with
                               probe Delta / 2); the `n_candidates` best rotations get two compass iterations each from the start shift
                               (first steps Delta / 2 and tol_shift / 2; bounds +-step about the grid rotation, +-tol_shift about the start
                               shift), and the best of them at the full band is refined again from Delta / 4 and tol_shift / 4 down to
                               step_tolerance; 2 = translation only (the protocol's mode 2).
                               Build-defined: the absent program ranks peaks of a spherical-harmonics correlation instead */
    float global_step;      /* degrees; 0 = 15 */
    int n_candidates;       /* metric/number_of_candidate_peaks_to_search; 0 = 25, at most 64 */
} ppm_sva_cfg;
/* volumes: n_vol * box^3 floats (x fastest); wedges: n_vol x {lwedge, uwedge} tilt limits in degrees (tilt axis = y);
 * poses: n_vol x 12 doubles {N row-major (9), shift x y z (pixels)}, start values in, refined values out; scores: n_vol. */
int ppm_sva_align(ppm_ref_t *ref, const ppm_sva_cfg *cfg, const void *volumes, int volumes_on_device, int n_vol, const float *wedges,
                  double *poses, double *scores);

/* symmetry: "C1", "Cn", "Dn", "T", "O", "I".  ext_device_buffer: NULL, or a device buffer of
 * ppm_accum_floats(box) floats the caller allocated (e.g. a torch tensor, so that RCCL can
 * reduce it in place); it must be zeroed by the caller. */
size_t ppm_accum_floats(int box);
ppm_accum_t *ppm_accum_create(int box, float pixel_size, const char *symmetry,
                              void *ext_device_buffer);
void ppm_accum_destroy(ppm_accum_t *acc);
int ppm_insert_batch(ppm_accum_t *acc, const ppm_recon_cfg *cfg, const void *images,
                     int images_on_device, int n_img, const double *rows);
/* host copies of the accumulators: ppm_accum_floats(box) floats laid out
 * [half 0..1][kz][ky][kx 0..box/2]{re, im, weight} */
int ppm_accum_download(ppm_accum_t *acc, float *host);
/* `count` floats from float index `first` of the same layout (one half map = ppm_accum_floats(box) / 2 floats): lets a caller
 * download each half into a page-locked buffer of its own and write the two dump files (frealign.py:1820-1822) from there */
int ppm_accum_download_range(ppm_accum_t *acc, float *host, size_t first, size_t count);
int ppm_accum_add(ppm_accum_t *acc, const float *host);
long ppm_accum_count(ppm_accum_t *acc, int half); /* particles inserted so far */
void ppm_accum_set_count(ppm_accum_t *acc, int half, long count);

/* Multi-GPU reconstruction: every rank (one process per GPU) inserts its particle shard into its own accumulator;
 * ONE sum over the ranks replaces the dump files + local_merge3d + the summation of merge3d (frealign.py:1838-1903, :2075-2093).
 * ppm_accum_reduce sums the accumulator (ppm_accum_floats(box) floats, in place on the device) and the two particle counters over
 * the communicator with RCCL on the library's stream: root >= 0 -> ncclReduce to that rank, root < 0 -> ncclAllReduce.  `comm` is
 * an ncclComm_t: the caller's own, or one made by ppm_comm_create (rank 0 calls ppm_comm_unique_id and hands the 128 bytes to the
 * other ranks by any means - MPI, a file, torch.distributed - then every rank calls ppm_comm_create; collective, blocks until all
 * ranks have joined).  librccl is opened on first use; if it cannot be, these calls fail with a message (no other transport). */
typedef struct ppm_comm_id { char bytes[128]; } ppm_comm_id; /* = ncclUniqueId */
int ppm_comm_unique_id(ppm_comm_id *id);
void *ppm_comm_create(int n_ranks, int rank, const ppm_comm_id *id);
int ppm_comm_count(void *comm); /* ranks of the communicator as RCCL reports them (ncclCommCount); < 0 on error */
void ppm_comm_destroy(void *comm);
int ppm_accum_reduce(ppm_accum_t *acc, void *comm, int root);

/* Sub-tomogram AVERAGE (BASELINE config 5 is "sub-tomogram averaging": a 3DAVG iteration ends in averaged maps,
 * `<dataset>_iteration_%03d_refined_selected_average_0.mrc` + `..._filtered.mrc`, src/pyp/refine/tomo_avg/sub_tomo_avg.py:79-94,
 * src/pyp_main.py:3076-3100; the next iteration aligns to them): the 3-D analogue of ppm_insert_batch.  Every sub-volume's transform
 * ((v - mean) / sigma, the whole box: no window, no band-pass) is brought into the reference frame by its aligned pose - the
 * convention of ppm_sva_align: F_v(k) = Ref(N k) e^{+2 pi i k.p / box} - and added to the accumulator of the half-map it belongs to
 * (parity of index[v], or of v when index is NULL; odd -> half 1), with its missing-wedge mask as the weight:
 *     num_h(q) += m_v(k) F_v(k) e^{-2 pi i k.p / box} / box,    den_h(q) += m_v(k),    k = N^T q  (trilinear interpolation of F_v)
 * for |q| < box/2 - 1, m_v = 1 where the tilt angle of (k_x, k_z) lies in the sub-volume's lwedge .. uwedge (cfg->use_missing_wedge)
 * and everywhere otherwise.  `acc` = ppm_accum_create(box, pixel, "C1", ...); ppm_accum_count counts sub-volumes; shards on several
 * GPUs are summed with ppm_accum_reduce; ppm_finalize turns the sums into the two half-maps, the FSC-weighted average and the
 * statistics table exactly as for a reconstruction.  Of cfg only box and use_missing_wedge are read.  Build-defined (the absent
 * MPI_Classification's weighting is not visible).  Sub-volumes enter the sums in batches of up to 32 and the counters follow every
 * completed batch: after an error return ppm_accum_count says how many are in the sums (the caller can go on from there or start
 * the accumulator again).  With index == NULL the half-map parity is that of the POSITION IN THIS CALL (plus the call's base in
 * ppm_sva_align_average): callers that split a table into several calls pass the table numbers in `index`. */
int ppm_sva_insert(ppm_accum_t *acc, const ppm_sva_cfg *cfg, const void *volumes, int volumes_on_device, int n_vol, const float *wedges,
                   const double *poses, const long *index);

/* One pass of a 3DAVG iteration: ppm_sva_align, and every chunk of sub-volumes is added to the average (ppm_sva_insert into `acc`) at
 * its refined pose while it is still in device memory - sub-volumes that live on the host (config 5: 10 k x 192^3 = 283 GB) cross PCIe
 * once per iteration instead of twice.  The calls on `acc` made inside run on the reference's stream: do not use `acc` from another
 * thread meanwhile. */
int ppm_sva_align_average(ppm_ref_t *ref, ppm_accum_t *acc, const ppm_sva_cfg *cfg, const void *volumes, int volumes_on_device, int n_vol,
                          const float *wedges, double *poses, double *scores, const long *index);

/* half1/half2/filtered: box^3 floats each (host).  stats: (box/2) * PPM_STATS_COLS doubles. */
int ppm_finalize(ppm_accum_t *acc, const ppm_final_cfg *cfg, float *half1, float *half2,
                 float *filtered, double *stats);

/* Particle extraction straight into a (resident) stack — the step before the path:
 * crop `box` x `box` windows around the picked coordinates, fill what falls outside the micrograph with the mean
 * of the inside part, replace empty boxes by white noise, subtract the background mean and divide by the background
 * sigma (pixels farther than radius_px from the box centre).  Restates extract_particles_non_mpi
 * (src/pyp/extract/core.py:447-506) and normalize_image / extract_background / fix_empty_particles_in_place
 * (src/pyp/analysis/image.py:320-340, :406-417, :461-471).
 * image: rows x cols floats (row-major); coords: m x 2 doubles {box[0] (column coordinate), box[1] (row coordinate)}
 * in unbinned pixels; out: m * box * box floats. */
int ppm_extract_boxes(const void *image, int image_on_device, int rows, int cols, const double *coords, int m,
                      int box, double coordinate_binning, double radius_px, int normalize, int fix_empty,
                      void *out, int out_on_device);

/* device-time accounting with HIP events on the library's stream */
void ppm_profile_enable(int on);
void ppm_profile_reset(void);
int ppm_profile_get(int kernel_id, double *total_ms, long *launches);

/* device memory helpers for callers that keep the particle stack resident in HBM */
void *ppm_device_alloc(size_t bytes);
void ppm_device_free(void *p);
int ppm_device_upload(void *dst, const void *src, size_t bytes);
int ppm_device_sync(void);
/* page-locked host staging memory: uploads from it overlap the kernels of the previous chunk (pageable memory makes them
 * synchronous); the drop-in executables read their particle ranges from the stack file through two such buffers */
void *ppm_host_alloc(size_t bytes);
void ppm_host_free(void *p);
/* Fill `dst` with `bytes` bytes of the open file `fd` from `offset`, read by `n_threads` (1..16) concurrent pread loops of a
 * thread pool the library keeps (no device call; returns when all parts are in).  The particle stack of a refine3d /
 * reconstruct3d range (src/pyp/refine/frealign/frealign.py:3918-3994, answer 1 "input particle images") comes out of the page
 * cache at ~60 GB/s this way, against ~8 GB/s for one thread.  Returns 0, -5 on a short read, -errno on a read error. */
int ppm_host_read(int fd, long long offset, void *dst, size_t bytes, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* PPM_H */
