"""Deterministic synthetic particle stacks with known poses.

Test/benchmark input generator, independent of both the HIP kernels and the CPU oracle: it
projects a phantom with torch tensor ops (works on CPU and on `cuda`) using a 2x zero-padded
Fourier volume and trilinear interpolation, applies the CTF, shifts, adds white noise and
normalises like the reference normalises extracted boxes (background outside the particle
radius -> mean 0 / sigma 1, src/pyp/analysis/image.py:406-417).  Rows are shaped like the
reference's from-scratch `.cistem` rows (src/pyp/inout/metadata/core.py:1324-1608).
"""
import math

import numpy as np
import torch

from .formats import cistem

SEED_VOLUME, SEED_POSES, SEED_NOISE = 20240501, 20240502, 20240503


def phantom(n, seed=SEED_VOLUME, n_blobs=40, n_atoms=None):
    """Protein-like phantom, (n,n,n) float32: `n_blobs` broad Gaussian blobs (domain-scale density)
    plus a cloud of point-like "atoms" blurred to ~1 px so that the spectrum does not die at high
    resolution, all inside radius 0.35 n, low-passed at 0.45 cycles/pixel."""
    rng = np.random.default_rng(seed)
    ax = np.arange(n, dtype=np.float32) - n // 2
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    v = np.zeros((n, n, n), dtype=np.float32)
    centres = []
    for _ in range(n_blobs):
        while True:
            c = rng.uniform(-0.35 * n, 0.35 * n, 3)
            if np.linalg.norm(c) <= 0.35 * n:
                break
        s = rng.uniform(3.0, 8.0) * n / 256.0
        s = max(s, 1.2)
        amp = rng.uniform(0.5, 1.5)
        centres.append((c, s))
        v += amp * np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (2 * s * s)).astype(np.float32)
    # atoms: scattered around the blob centres, nearest-voxel deposition, Gaussian blur in Fourier space
    if n_atoms is None:
        n_atoms = 60 * n
    a = np.zeros((n, n, n), dtype=np.float32)
    which = rng.integers(0, n_blobs, n_atoms)
    for i in range(n_atoms):
        c, s = centres[which[i]]
        p = c + rng.normal(0, 1.5 * s, 3)
        if np.linalg.norm(p) > 0.35 * n:
            continue
        q = np.rint(p).astype(int) + n // 2
        a[q[2], q[1], q[0]] += rng.uniform(0.5, 1.5)
    k = np.fft.fftfreq(n)
    kz, ky, kx = np.meshgrid(k, k, k, indexing="ij")
    k2 = kx ** 2 + ky ** 2 + kz ** 2
    f = np.fft.fftn(v) + 3.0 * np.fft.fftn(a) * np.exp(-2 * (np.pi ** 2) * k2 * 0.9 ** 2)
    f[np.sqrt(k2) > 0.45] = 0
    return np.real(np.fft.ifftn(f)).astype(np.float32)


def phantom_sym(n, ops, seed=SEED_VOLUME, n_blobs=12):
    """Phantom with the point-group symmetry given by `ops` (k x 3 x 3 rotation matrices acting on (x, y, z)): Gaussian blobs
    (sigma 1.2-3 px at n = 64, scaled with n) replicated under every operator, inside radius 0.3 n."""
    rng = np.random.default_rng(seed)
    ax = np.arange(n, dtype=np.float32) - n // 2
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    v = np.zeros((n, n, n), dtype=np.float32)
    for _ in range(n_blobs):
        while True:
            c = rng.uniform(-0.3 * n, 0.3 * n, 3)
            if 0.08 * n <= np.linalg.norm(c) <= 0.3 * n:
                break
        s = max(1.2, rng.uniform(1.2, 3.0) * n / 64.0)
        amp = rng.uniform(0.5, 1.5)
        for R in np.asarray(ops, dtype=np.float64).reshape(-1, 3, 3):
            q = R @ c
            v += amp * np.exp(-((x - q[0]) ** 2 + (y - q[1]) ** 2 + (z - q[2]) ** 2) / (2 * s * s)).astype(np.float32)
    return v


def euler_matrix(psi, theta, phi):
    """M = Rz(phi) Ry(theta) Rz(psi) (degrees): image-plane coordinates -> reference coordinates
    ("rotates the reference by PHI -> THETA -> PSI", src/pyp/analysis/geometry/core.py:1186-1187)."""
    ps, th, ph = np.radians(psi), np.radians(theta), np.radians(phi)
    c, s = np.cos, np.sin
    return np.array([
        [c(ph) * c(th) * c(ps) - s(ph) * s(ps), -c(ph) * c(th) * s(ps) - s(ph) * c(ps), c(ph) * s(th)],
        [s(ph) * c(th) * c(ps) + c(ph) * s(ps), -s(ph) * c(th) * s(ps) + c(ph) * c(ps), s(ph) * s(th)],
        [-s(th) * c(ps), s(th) * s(ps), c(th)]])


def pyp_matrix(psi, theta, phi):
    """The matrix PYP writes out at analysis/geometry/core.py:1194-1197 (its "left-handed" form).  It equals
    euler_matrix(-psi, -theta, -phi); kept for the convention golden test."""
    ps, th, ph = np.radians(psi), np.radians(theta), np.radians(phi)
    c, s = np.cos, np.sin
    return np.array([
        [c(ph) * c(th) * c(ps) - s(ph) * s(ps), c(ph) * c(th) * s(ps) + s(ph) * c(ps), -c(ph) * s(th)],
        [-s(ph) * c(th) * c(ps) - c(ph) * s(ps), -s(ph) * c(th) * s(ps) + c(ph) * c(ps), s(ph) * s(th)],
        [s(th) * c(ps), s(th) * s(ps), c(th)]])


def angles_from_pyp_matrix(m):
    """(psi, theta, phi) in [0,360) from the matrix above; restates get_degrees_from_matrix
    (analysis/geometry/core.py:211-234)."""
    eps = np.nextafter(0, 1)
    if m[2, 2] < 1 - eps:
        if m[2, 2] > -1 + eps:
            theta = math.acos(m[2, 2])
            st = math.sin(theta)
            psi = math.atan2(m[2, 1] / st, m[2, 0] / st)
            phi = math.atan2(m[1, 2] / st, -m[0, 2] / st)
        else:
            theta, phi, psi = math.pi, math.atan2(-m[0, 1], -m[0, 0]), 0.0
    else:
        theta, phi, psi = 0.0, math.atan2(m[0, 1], m[0, 0]), 0.0
    out = np.degrees([psi, theta, phi])
    return tuple(np.where(out < 0, out + 360.0, out))


def ctf_image(n, pixel, df1, df2, angast_deg, kv, cs_mm, amp, phase_shift, device):
    """CTF on the centred full grid (ky, kx in -n/2..n/2-1); tensors of shape (M,) broadcast to (M,n,n)."""
    k = torch.arange(-n // 2, n // 2, device=device, dtype=torch.float32)
    ky, kx = torch.meshgrid(k, k, indexing="ij")
    v = kv * 1000.0
    lam = 12.2639 / math.sqrt(v + 0.97845e-6 * v * v)
    s2 = (kx * kx + ky * ky) / (n * pixel) ** 2
    ang = torch.atan2(ky, kx)
    df1, df2, ast = (torch.as_tensor(t, device=device, dtype=torch.float32).view(-1, 1, 1) for t in (df1, df2, np.radians(angast_deg)))
    df = 0.5 * (df1 + df2 + (df1 - df2) * torch.cos(2 * (ang - ast)))
    chi = math.pi * lam * s2 * (df - 0.5 * cs_mm * 1e7 * lam * lam * s2) + phase_shift + math.atan(amp / math.sqrt(1 - amp * amp))
    return -torch.sin(chi)


class Projector:
    """Fourier-slice projector over a 2x padded volume (torch)."""

    def __init__(self, vol, device="cpu"):
        self.n = n = vol.shape[0]
        self.device = torch.device(device)
        p = 2 * n
        v = torch.zeros((p, p, p), dtype=torch.float32, device=self.device)
        o = (p - n) // 2
        v[o:o + n, o:o + n, o:o + n] = torch.as_tensor(vol, device=self.device)
        f = torch.fft.fftshift(torch.fft.fftn(torch.fft.ifftshift(v)))      # centred spectrum, origin at p/2
        self.f = f.to(torch.complex64)
        self.p = p
        k = torch.arange(-n // 2, n // 2, device=self.device, dtype=torch.float32)
        self.ky, self.kx = torch.meshgrid(k, k, indexing="ij")

    def spectra(self, mats):
        """mats: (M,3,3) rotation matrices -> (M,n,n) complex centred projection spectra (unnormalised FFT2)."""
        m = torch.as_tensor(np.asarray(mats), device=self.device, dtype=torch.float32)
        p, f = self.p, self.f
        kx, ky = self.kx[None], self.ky[None]
        X = 2 * (m[:, 0, 0, None, None] * kx + m[:, 0, 1, None, None] * ky) + p // 2
        Y = 2 * (m[:, 1, 0, None, None] * kx + m[:, 1, 1, None, None] * ky) + p // 2
        Z = 2 * (m[:, 2, 0, None, None] * kx + m[:, 2, 1, None, None] * ky) + p // 2
        x0, y0, z0 = X.floor(), Y.floor(), Z.floor()
        fx, fy, fz = X - x0, Y - y0, Z - z0
        out = torch.zeros(X.shape, dtype=torch.complex64, device=self.device)
        flat = f.reshape(-1)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    xi, yi, zi = (x0 + dx).long(), (y0 + dy).long(), (z0 + dz).long()
                    ok = (xi >= 0) & (xi < p) & (yi >= 0) & (yi < p) & (zi >= 0) & (zi < p)
                    idx = (zi.clamp(0, p - 1) * p + yi.clamp(0, p - 1)) * p + xi.clamp(0, p - 1)
                    w = (fx if dx else 1 - fx) * (fy if dy else 1 - fy) * (fz if dz else 1 - fz)
                    out += torch.where(ok, w, torch.zeros_like(w)) * flat[idx]
        return out


def make_dataset(n, m, pixel=1.0, snr=0.05, seed_poses=SEED_POSES, seed_noise=SEED_NOISE, vol=None,
                 device="cpu", batch=64, particle_rad_frac=0.32, shift_sigma=2.0, shift_clip=6.0,
                 kv=300.0, cs_mm=2.7, amp=0.07, unique=None, normalize=True):
    """Return (vol, stack (m,n,n) float32 torch tensor on `device`, rows (m,32) float64 with the TRUE poses).

    unique: if set (< m), only that many distinct clean projections are computed and reused cyclically with
    fresh noise — used by bench.py to fill a 100k-particle stack quickly; every particle is still a distinct image."""
    if vol is None:
        vol = phantom(n)
    rng = np.random.default_rng(seed_poses)
    u = m if unique is None else min(unique, m)
    psi, phi = rng.uniform(0, 360, u), rng.uniform(0, 360, u)
    theta = np.degrees(np.arccos(rng.uniform(-1, 1, u)))
    sh = np.clip(rng.normal(0, shift_sigma, (u, 2)), -shift_clip, shift_clip)
    df1 = rng.uniform(8000, 24000, u)
    df2 = df1 + rng.uniform(-300, 300, u)
    ast = rng.uniform(0, 180, u)
    rows = cistem.default_rows(m, pixel, kv, cs_mm, amp)
    rep = np.arange(m) % u
    C = cistem.COL
    rows[:, C["PSI"]], rows[:, C["THETA"]], rows[:, C["PHI"]] = psi[rep], theta[rep], phi[rep]
    rows[:, C["X_SHIFT"]], rows[:, C["Y_SHIFT"]] = sh[rep, 0] * pixel, sh[rep, 1] * pixel
    rows[:, C["DEFOCUS_1"]], rows[:, C["DEFOCUS_2"]], rows[:, C["DEFOCUS_ANGLE"]] = df1[rep], df2[rep], ast[rep]

    stack = render_rows(vol, rows[:u], pixel, snr, seed_noise, device, batch, particle_rad_frac, kv, cs_mm, amp, m=m, rep=rep, normalize=normalize)
    return vol, stack, rows


def render_rows(vol, rows, pixel=1.0, snr=0.05, seed_noise=SEED_NOISE, device="cpu", batch=64, particle_rad_frac=0.32,
                kv=300.0, cs_mm=2.7, amp=0.07, m=None, rep=None, normalize=True):
    """Images of the poses / CTF parameters in `rows` (u x 32): projection x CTF, shifted, white noise at `snr`, background
    normalised (unless normalize is False: the images keep the reference's density scale).  With m / rep, image i of the m returned is clean projection rep[i] with fresh noise."""
    C = cistem.COL
    u = len(rows)
    n = vol.shape[0]
    if m is None:
        m, rep = u, np.arange(u)
    psi, theta, phi = rows[:, C["PSI"]], rows[:, C["THETA"]], rows[:, C["PHI"]]
    sh = np.stack([rows[:, C["X_SHIFT"]], rows[:, C["Y_SHIFT"]]], axis=1) / pixel
    df1, df2, ast = rows[:, C["DEFOCUS_1"]], rows[:, C["DEFOCUS_2"]], rows[:, C["DEFOCUS_ANGLE"]]
    dev = torch.device(device)
    proj = Projector(vol, dev)
    k = torch.arange(-n // 2, n // 2, device=dev, dtype=torch.float32)
    ky, kx = torch.meshgrid(k, k, indexing="ij")
    clean = torch.empty((u, n, n), dtype=torch.float32, device=dev)
    for b0 in range(0, u, batch):
        b1 = min(u, b0 + batch)
        mats = np.stack([euler_matrix(psi[i], theta[i], phi[i]) for i in range(b0, b1)])
        spec = proj.spectra(mats)
        ctf = ctf_image(n, pixel, df1[b0:b1], df2[b0:b1], ast[b0:b1], kv, cs_mm, amp, 0.0, dev)
        sx = torch.as_tensor(sh[b0:b1, 0], device=dev, dtype=torch.float32).view(-1, 1, 1)
        sy = torch.as_tensor(sh[b0:b1, 1], device=dev, dtype=torch.float32).view(-1, 1, 1)
        ramp = torch.exp(-2j * math.pi * (kx[None] * sx + ky[None] * sy) / n)
        spec = spec * ctf * ramp
        bt = rows[b0:b1][:, [C["BEAM_TILT_X"], C["BEAM_TILT_Y"]]]
        if np.any(bt != 0):     # tilted beam (mrad): the image transform is multiplied by exp(+i phi), phi = 2 pi Cs lambda^2 |s|^2 (s . b)
            v = kv * 1000.0
            lam = 12.2639 / math.sqrt(v + 0.97845e-6 * v * v)
            cc = 2.0 * math.pi * cs_mm * 1e7 * lam * lam * 1e-3 / (n * pixel) ** 3
            bx = torch.as_tensor(bt[:, 0] * cc, device=dev, dtype=torch.float32).view(-1, 1, 1)
            by = torch.as_tensor(bt[:, 1] * cc, device=dev, dtype=torch.float32).view(-1, 1, 1)
            spec = spec * torch.exp(1j * (kx * kx + ky * ky)[None] * (kx[None] * bx + ky[None] * by))
        img = torch.fft.fftshift(torch.fft.ifft2(torch.fft.ifftshift(spec, dim=(-2, -1))), dim=(-2, -1)).real
        clean[b0:b1] = img
    sig_var = clean.var(dim=(-2, -1), keepdim=True).mean()
    noise_sd = float(torch.sqrt(sig_var / snr)) if snr > 0 else 0.0
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed_noise)
    stack = torch.empty((m, n, n), dtype=torch.float32, device=dev)
    rad = particle_rad_frac * n
    bg = ((kx * kx + ky * ky) > rad * rad)
    for b0 in range(0, m, 1024):
        b1 = min(m, b0 + 1024)
        idx = torch.as_tensor(rep[b0:b1], device=dev)
        x = clean[idx]
        if noise_sd > 0:
            x = x + noise_sd * torch.randn(x.shape, generator=gen, device=dev)
        # cryo-EM convention: protein is dark -> invert? we keep positive-density-is-positive and
        # the callers pass invert = no.
        mu = x[:, bg].mean(dim=1).view(-1, 1, 1)
        sd = x[:, bg].std(dim=1, unbiased=False).view(-1, 1, 1)
        stack[b0:b1] = (x - mu) / sd if normalize else x
    return stack




def rot_xyz(k, deg):
    """Right-handed rotation about x (0), y (1), z (2)."""
    t = np.radians(deg)
    c, s = np.cos(t), np.sin(t)
    return np.array([[[1, 0, 0], [0, c, -s], [0, s, c]], [[c, 0, s], [0, 1, 0], [-s, 0, c]], [[c, -s, 0], [s, c, 0], [0, 0, 1]]][k], dtype=np.float64)


def angles_from_matrix(M):
    """(psi, theta, phi) in [0, 360) of M = euler_matrix(psi, theta, phi)."""
    ct = min(1.0, max(-1.0, M[2, 2]))
    st = math.hypot(M[0, 2], M[1, 2])
    if st > 1e-7:
        theta, phi, psi = math.atan2(st, ct), math.atan2(M[1, 2], M[0, 2]), math.atan2(M[2, 1], -M[2, 0])
    else:
        theta, phi = (0.0 if ct > 0 else math.pi), 0.0
        psi = math.atan2(M[1, 0], M[0, 0]) if ct > 0 else math.atan2(-M[1, 0], -M[0, 0])
    out = np.degrees([psi, theta, phi])
    out[out < 0] += 360.0
    return out


def csp_row_pose(particle, tilt):
    """Pose of a projection row from its particle (12 columns of the particle block) and tilt (6 columns of the tilt block):
    returns (psi, theta, phi, gx, gy) with the geometric shift g in pixels.  Numpy statement of the relation in
    include/ppm.h (ppm_csp_cfg), which restates csp_euler_angles (src/pyp/analysis/geometry/core.py:1081-1213)."""
    N = euler_matrix(-particle[4], -particle[5], -particle[6])
    M = N @ rot_xyz(1, -tilt[4]) @ rot_xyz(2, tilt[5])
    g = rot_xyz(2, -tilt[5]) @ rot_xyz(1, tilt[4]) @ (-np.asarray(particle[1:4], dtype=np.float64))
    a = angles_from_matrix(M)
    return np.array([a[0], a[1], a[2], g[0] + tilt[2], g[1] + tilt[3]])


def make_tilt_series(n, n_part, tilt_angles, pixel=1.0, snr=0.1, vol=None, seed=20240601, device="cpu", axis=85.0,
                     particle_rad_frac=0.32, defocus=(15000.0, 25000.0), dose_defocus_slope=0.0):
    """Synthetic constrained data set (one tilt series): `n_part` particles with random 3-D orientations and small 3-D shifts,
    one projection per (particle, tilt).  Returns (vol, stack (n_part * n_tilt, n, n), rows (M, 32), particles (n_part, 12),
    tilts (n_tilt, 6)): the TRUE parameters; rows carry the poses that follow from them (csp_row_pose) and a random
    sub-pixel residual in X_SHIFT / Y_SHIFT like the reference's extraction leaves (metadata/core.py:2745-2752)."""
    if vol is None:
        vol = phantom(n)
    rng = np.random.default_rng(seed)
    n_tilt = len(tilt_angles)
    particles = np.zeros((n_part, 12))
    particles[:, 0] = np.arange(n_part)
    particles[:, 1:4] = rng.normal(0, 1.0, (n_part, 3))
    for i in range(n_part):
        a = angles_from_matrix(euler_matrix(rng.uniform(0, 360), np.degrees(np.arccos(rng.uniform(-1, 1))), rng.uniform(0, 360)))
        particles[i, 4:7] = -a
    particles[:, 7:10] = rng.uniform(100, 900, (n_part, 3))
    particles[:, 10], particles[:, 11] = 0.0, 100.0
    tilts = np.zeros((n_tilt, 6))
    tilts[:, 0] = np.arange(n_tilt)
    tilts[:, 2:4] = rng.normal(0, 0.5, (n_tilt, 2))
    tilts[:, 4] = np.asarray(tilt_angles, dtype=np.float64)
    tilts[:, 5] = axis + rng.normal(0, 0.3, n_tilt)
    m = n_part * n_tilt
    rows = cistem.default_rows(m, pixel, 300.0, 2.7, 0.07)
    C = cistem.COL
    resid = rng.uniform(-0.5, 0.5, (m, 2))
    df = rng.uniform(defocus[0], defocus[1], n_tilt)
    j = 0
    for ip in range(n_part):
        for it in range(n_tilt):
            pose = csp_row_pose(particles[ip], tilts[it])
            rows[j, C["PSI"]], rows[j, C["THETA"]], rows[j, C["PHI"]] = pose[0], pose[1], pose[2]
            rows[j, C["X_SHIFT"]], rows[j, C["Y_SHIFT"]] = (pose[3] + resid[j, 0]) * pixel, (pose[4] + resid[j, 1]) * pixel
            rows[j, C["DEFOCUS_1"]] = df[it] + 40.0 * (ip % 7)
            rows[j, C["DEFOCUS_2"]] = rows[j, C["DEFOCUS_1"]] - 200.0
            rows[j, C["DEFOCUS_ANGLE"]] = 30.0
            rows[j, C["PIND"]], rows[j, C["TIND"]], rows[j, C["IMIND"]] = ip, it, it
            j += 1
    stack = render_rows(vol, rows, pixel, snr, seed + 1, device, 64, particle_rad_frac)
    return vol, stack, rows, particles, tilts


def make_subtomograms(n, n_vol, pixel=1.0, snr=0.5, vol=None, seed=20240701, wedge=(-60.0, 60.0), shift_sigma=1.5, device="cpu"):
    """Synthetic sub-tomograms of `vol`: sub-volume v has the transform ref(N_v k) e^{+2 pi i k.p_v / n} inside the measured
    wedge (tilt axis y, tilt range `wedge` degrees), zero in the missing wedge, plus white noise.  Returns (vol, volumes
    (n_vol, n, n, n) float32, poses (n_vol, 12) = N row-major + shift, wedges (n_vol, 2))."""
    if vol is None:
        vol = phantom(n)
    rng = np.random.default_rng(seed)
    dev = torch.device(device)
    proj = Projector(vol, dev)                  # 2 x padded centred transform
    p2 = proj.p
    k = torch.arange(-n // 2, n // 2, device=dev, dtype=torch.float32)
    kz, ky, kx = torch.meshgrid(k, k, k, indexing="ij")
    poses = np.zeros((n_vol, 12))
    out = torch.empty((n_vol, n, n, n), dtype=torch.float32, device=dev)
    ang = torch.rad2deg(torch.atan2(kz, kx))
    ang = torch.where(ang > 90, ang - 180, ang)
    ang = torch.where(ang <= -90, ang + 180, ang)
    inw = ((ang >= wedge[0]) & (ang <= wedge[1])) | ((kx == 0) & (kz == 0))
    band = (kx * kx + ky * ky + kz * kz) < (n / 2 - 1) ** 2
    flat = proj.f.reshape(-1)
    for v in range(n_vol):
        Nm = euler_matrix(rng.uniform(0, 360), np.degrees(np.arccos(rng.uniform(-1, 1))), rng.uniform(0, 360))
        p = rng.normal(0, shift_sigma, 3)
        poses[v, :9], poses[v, 9:] = Nm.ravel(), p
        m = torch.as_tensor(Nm, device=dev, dtype=torch.float32)
        X = 2 * (m[0, 0] * kx + m[0, 1] * ky + m[0, 2] * kz) + p2 // 2
        Y = 2 * (m[1, 0] * kx + m[1, 1] * ky + m[1, 2] * kz) + p2 // 2
        Z = 2 * (m[2, 0] * kx + m[2, 1] * ky + m[2, 2] * kz) + p2 // 2
        x0, y0, z0 = X.floor(), Y.floor(), Z.floor()
        fx, fy, fz = X - x0, Y - y0, Z - z0
        spec = torch.zeros(X.shape, dtype=torch.complex64, device=dev)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    xi, yi, zi = (x0 + dx).long(), (y0 + dy).long(), (z0 + dz).long()
                    ok = (xi >= 0) & (xi < p2) & (yi >= 0) & (yi < p2) & (zi >= 0) & (zi < p2)
                    idx = (zi.clamp(0, p2 - 1) * p2 + yi.clamp(0, p2 - 1)) * p2 + xi.clamp(0, p2 - 1)
                    w = (fx if dx else 1 - fx) * (fy if dy else 1 - fy) * (fz if dz else 1 - fz)
                    spec += torch.where(ok, w, torch.zeros_like(w)) * flat[idx]
        ramp = torch.exp(2j * math.pi * (kx * float(p[0]) + ky * float(p[1]) + kz * float(p[2])) / n)
        spec = spec * ramp * (inw & band)
        real = torch.fft.fftshift(torch.fft.ifftn(torch.fft.ifftshift(spec))).real
        out[v] = real
    sig = out.var(dim=(1, 2, 3)).mean()
    if snr > 0:
        gen = torch.Generator(device=dev); gen.manual_seed(seed + 1)
        out = out + float(torch.sqrt(sig / snr)) * torch.randn(out.shape, generator=gen, device=dev)
    wedges = np.tile(np.asarray(wedge, dtype=np.float32), (n_vol, 1))
    return vol, out, poses, wedges


def perturb_poses(poses, angle_sigma=3.0, shift_sigma=1.0, seed=11):
    """Copies of (N, shift) poses with small random rotations about the specimen axes and shifts added."""
    rng = np.random.default_rng(seed)
    out = poses.copy()
    for v in range(len(out)):
        Nm = out[v, :9].reshape(3, 3)
        for k in range(3):
            Nm = Nm @ rot_xyz(k, rng.normal(0, angle_sigma))
        out[v, :9] = Nm.ravel()
        out[v, 9:] += rng.normal(0, shift_sigma, 3)
    return out


def pose_angle_error(a, b):
    """Rotation angle (degrees) between the N matrices of two pose tables."""
    out = []
    for x, y in zip(a, b):
        t = (np.trace(x[:9].reshape(3, 3).T @ y[:9].reshape(3, 3)) - 1.0) / 2.0
        out.append(np.degrees(np.arccos(np.clip(t, -1.0, 1.0))))
    return np.array(out)


def paste_tilt_series(stack, rows, n_tilt, shape=(512, 512), seed=5):
    """Tilt-series images (n_tilt, H, W) with every projection of `stack` pasted at a well-separated integer position of its
    tilt image (IMIND) on a unit-noise background; sets ORIGINAL_X_POSITION (column) / ORIGINAL_Y_POSITION (row) in `rows`
    (returned copy) so that extracting a box of the same size around them gives the projection back."""
    C = cistem.COL
    on_gpu = hasattr(stack, "is_cuda") and stack.is_cuda          # a resident stack: the series is built on its device (41 x 4096^2 is 2.7 GB)
    st = stack if on_gpu else (stack.numpy() if hasattr(stack, "numpy") else np.asarray(stack))
    m, n = st.shape[0], st.shape[1]
    rng = np.random.default_rng(seed)
    if on_gpu:
        gen = torch.Generator(device=stack.device); gen.manual_seed(seed)
        series = torch.randn((n_tilt,) + tuple(shape), generator=gen, device=stack.device, dtype=torch.float32)
    else:
        series = rng.normal(0, 1, (n_tilt,) + tuple(shape)).astype(np.float32)
    rows = rows.copy()
    pind = np.unique(rows[:, C["PIND"]].astype(int))
    per_row = max(1, (shape[1] - n) // (n + 16))
    centre = {}
    for k, pid in enumerate(pind):
        gy, gx = divmod(k, per_row)
        centre[pid] = (n // 2 + 8 + gx * (n + 16), n // 2 + 8 + gy * (n + 16))
        if centre[pid][1] + n // 2 >= shape[0]:
            raise ValueError("ERROR: tilt image too small for the particles")
    for j in range(m):
        x, y = centre[int(rows[j, C["PIND"]])]
        x, y = x + int(rng.integers(-3, 4)), y + int(rng.integers(-3, 4))
        t = int(rows[j, C["IMIND"]])
        series[t, y - n // 2:y - n // 2 + n, x - n // 2:x - n // 2 + n] = st[j]
        rows[j, C["ORIGINAL_X_POSITION"]], rows[j, C["ORIGINAL_Y_POSITION"]] = x, y
    return series, rows


def csp_rows_from_params(rows_ref, particles_ref, tilts_ref, particles, tilts):
    """Rows that follow from (particles, tilts) when `rows_ref` followed from (particles_ref, tilts_ref): angles from the
    geometry, shifts moved by the change of the geometric shift (what ppm_csp_refine writes back)."""
    C = cistem.COL
    out = rows_ref.copy()
    pidx = {int(p[0]): i for i, p in enumerate(particles)}
    tidx = {(int(t[0]), int(t[1])): i for i, t in enumerate(tilts)}
    for j, r in enumerate(rows_ref):
        ip, it = pidx[int(r[C["PIND"]])], tidx[(int(r[C["TIND"]]), int(r[C["RIND"]]))]
        a, b = csp_row_pose(particles_ref[ip], tilts_ref[it]), csp_row_pose(particles[ip], tilts[it])
        px = r[C["PIXEL_SIZE"]]
        out[j, C["PSI"]], out[j, C["THETA"]], out[j, C["PHI"]] = b[0], b[1], b[2]
        out[j, C["X_SHIFT"]] = r[C["X_SHIFT"]] + (b[3] - a[3]) * px
        out[j, C["Y_SHIFT"]] = r[C["Y_SHIFT"]] + (b[4] - a[4]) * px
    return out


def perturb_rows(rows, angle_sigma=2.0, shift_sigma_px=1.0, pixel=1.0, seed=7):
    """Copy of `rows` with Gaussian perturbations of the poses (config 1: local refinement start)."""
    rng = np.random.default_rng(seed)
    r = rows.copy()
    C = cistem.COL
    for c in ("PSI", "THETA", "PHI"):
        r[:, C[c]] += rng.normal(0, angle_sigma, len(r))
    for c in ("X_SHIFT", "Y_SHIFT"):
        r[:, C[c]] += rng.normal(0, shift_sigma_px * pixel, len(r))
    return r


def angular_error_deg(rows_a, rows_b, ops=None):
    """Geodesic angle (degrees) between the rotations of two row sets; with `ops` (k x 3 x 3 point-group operators acting on
    the reference frame) the smallest angle over the symmetry-equivalent poses S M."""
    C = cistem.COL
    out = np.empty(len(rows_a))
    sym = [np.eye(3)] if ops is None else list(np.asarray(ops, dtype=np.float64).reshape(-1, 3, 3))
    for i in range(len(rows_a)):
        ma = euler_matrix(rows_a[i, C["PSI"]], rows_a[i, C["THETA"]], rows_a[i, C["PHI"]])
        mb = euler_matrix(rows_b[i, C["PSI"]], rows_b[i, C["THETA"]], rows_b[i, C["PHI"]])
        t = max((np.trace((S @ ma).T @ mb) - 1.0) / 2.0 for S in sym)
        out[i] = np.degrees(np.arccos(np.clip(t, -1.0, 1.0)))
    return out


def shift_error_px(rows_a, rows_b, pixel=1.0):
    C = cistem.COL
    d = rows_a[:, [C["X_SHIFT"], C["Y_SHIFT"]]] - rows_b[:, [C["X_SHIFT"], C["Y_SHIFT"]]]
    return np.sqrt((d ** 2).sum(axis=1)) / pixel
