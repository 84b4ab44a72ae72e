#!/usr/bin/env python3
"""bench.py — particles/s of per-particle projection matching (BASELINE.json metric) on N MI355X.

A "step" is one pass of the hot path (pre-processing FFT + CTF tables, global grid search,
top-hit + final local refinement: one ppm_refine_batch call) over the rank's synthetic
particle stack, which is resident in HBM before the timed region starts.

Workload at N=1 = BASELINE.json configs[1]: "SPA global search: 100k 256^2 particles,
15 deg angular step, 1 MI355X" (search band r = 64 Fourier px).  Particles
shard across ranks with no data-path collective (weak scaling: --particles is per GPU).
The same line carries a second block, "reconstruct" = configs[2]: Fourier insertion of
500k 256^2 particles per GPU into the half-map accumulators + one all-reduce (RCCL).

Contract: python bench.py --gpus N --steps K --warmup W
  * N > 1 without a torchrun environment: this process starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`
    as a CHILD (it never touches the GPU itself) and exits with the child's code;
  * under torchrun (RANK / LOCAL_RANK / WORLD_SIZE set) it is one rank of N.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_VALU_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak fp32 vector
PEAK_HBM_GBPS = 8000.0        # HBM3E spec
PEAK_L1_NOTE = "derived, not a figure of MI355X_MICROARCH.md: 64 B per clock and CU x 256 CUs x 2.4 GHz"
PEAK_L1_GBPS = 64.0 * 256 * 2.4   # vector L1 delivery, 64 B per clock and CU at the 2.4 GHz peak clock = 39.3 TB/s (the gather kernels measure
                                  # 56 B per clock with every load an L1 hit, and the chip sustains ~1.95 GHz under them)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--particles", type=int, default=100000, help="particles per GPU per step of the refinement workload (resident stack)")
    ap.add_argument("--recon-particles", type=int, default=500000, help="particles per GPU per step of the reconstruction block")
    ap.add_argument("--box", type=int, default=256)
    ap.add_argument("--band", type=float, default=64.0, help="search / refinement band limit, Fourier pixels")
    ap.add_argument("--angular-step", type=float, default=15.0)
    ap.add_argument("--search-range", type=float, default=6.0, help="shift search range of the grid search, pixels (BASELINE config 2: shifts clipped at +-6 px); 0 = the mask radius, what PYP's default refine_searchx = 0 asks for (full-window transform kernel k_gfft; the default line reports it as the `default_search` block)")
    ap.add_argument("--unique", type=int, default=0, help="distinct clean projections (0 = one per particle: every particle has its own pose)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of each CPU oracle leg (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", choices=["both", "refine", "reconstruct", "csp", "sva"], default="both",
                    help="both = the refinement line (BASELINE.json configs[1], the headline metric) carrying a 'reconstruct' block "
                         "(configs[2]) and the two next-row blocks 'csp' (configs[3]: constrained tilt-series refinement) and 'sva' "
                         "(configs[4]: sub-tomogram alignment); refine / reconstruct / csp / sva = that workload alone")
    ap.add_argument("--csp-particles", type=int, default=500, help="particles of the tilt series of the csp block (41 tilts, 128^2 boxes)")
    ap.add_argument("--sva-volumes", type=int, default=512, help="resident 192^3 sub-volumes of the sva block")
    ap.add_argument("--no-next-rows", action="store_true", help="leave the csp / sva blocks out of the default line")
    ap.add_argument("--no-side", action="store_true", help="csp / sva blocks: only the timed figure (no second series in flight, no global search, no average) - what the "
                                                           "counter passes of scripts/pmc_r04.sh profile, so that a kernel's dispatches are those of the timed calls")
    ap.add_argument("--no-dropin", action="store_true", help="leave the 'dropin' block (bin/refine3d + bin/reconstruct3d as child processes on a stack file) out")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------- launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a, argv):
    """--gpus N outside torchrun: start the N ranks as a child torchrun.  Nothing here initialises the GPU
    (torch.cuda.device_count() does not), and the child is a new process, not an exec of this one."""
    probe = os.environ.get("PPM_BENCH_PROBE") == "1"
    if not probe and "PPM_FORCE_DEVICE" not in os.environ:
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            print(f"ERROR: bench.py --gpus {a.gpus}: only {have} GPU(s) visible", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def probe_main(a, rank, world):
    """PPM_BENCH_PROBE=1: the launcher / rendezvous / timing-reduction path without any GPU work (CPU test of --gpus N)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(os.environ.get("PPM_DIST_BACKEND", "gloo"))
        dist.barrier()
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "particles/sec projection-matching, 256^2 box", "value": None, "unit": "particles/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "probe": True, "max_over_ranks_s": float(t.item())}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# --------------------------------------------------------------------------------------------- helpers
KERNEL_SOURCES = ("ppm_dev.h", "ppm_kernels.h", "ppm_kernels2.h", "ppm_csp_kernels.h", "ppm_sva_kernels.h", "ppm_gfft.h", "ppm_fft_reg.h")


def so_sha16():
    import hashlib
    p = os.path.join(ROOT, "pyp_amd", "libpypmatch.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16] if os.path.exists(p) else None


KERNEL_SOURCES_MAIN = ("ppm_dev.h", "ppm_kernels.h", "ppm_kernels2.h", "ppm_gfft.h", "ppm_fft_reg.h")      # the refinement and insertion kernels (what the PMC summaries profile)


def kernels_sha16(files=KERNEL_SOURCES):
    """Identity of the DEVICE code (the kernel headers): a PMC summary stays valid across host-only changes of the library."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "pyp_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_latest(workload):
    """The newest committed PMC summary of a workload (profiles/rNN_pmc_<workload>.json)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_pmc_%s.json" % workload)))
    return os.path.basename(c[-1]) if c else "r02_pmc_%s.json" % workload


def pmc_entry(summary, kernel, merge=False):
    """One kernel's counters from a committed rocprofv3 --pmc summary (scripts/pmc_r02.sh -> scripts/pmc_summary.py).
    Returns (entry, meta) or (None, None)."""
    path = os.path.join(ROOT, "profiles", summary)
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    meta = d.get("_meta", {})
    key = [k for k in d if kernel in k]
    if len(key) > 1 and merge:            # template instances of one kernel (k_sva_eval<0>, k_sva_eval<6>): counters added up
        out = {}
        for k in key:
            for c, v in d[k].items():
                o = out.setdefault(c, {"sum": 0.0, "dispatches": 0})
                o["sum"] += v["sum"]; o["dispatches"] += v["dispatches"]
        for v in out.values():
            v["per_dispatch"] = v["sum"] / max(v["dispatches"], 1)
        return out, meta
    return (d[key[0]], meta) if key else (None, meta)


def pmc_traffic(summary, kernel, particles_per_launch):
    """HBM-side bytes per launch of `kernel`: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes (units of 1 KB);
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 B); scaled from the
    profiled particle count to this launch.  None when no summary with these counters is committed; the source string says
    whether the summary was taken from the library build that is being timed."""
    e, meta = pmc_entry(summary, kernel)
    if not e or "FETCH_SIZE" not in e or "WRITE_SIZE" not in e or not meta.get("particles"):
        return None, "no FETCH_SIZE / WRITE_SIZE summary for %s under profiles/%s" % (kernel, summary)
    per_particle = (2.0 * e["FETCH_SIZE"]["sum"] + e["WRITE_SIZE"]["sum"]) * 1024.0 / meta["particles"]
    # compared on the headers of the refinement / insertion kernels when the summary names them (changes of the csp / sva kernels
    # do not make these counters stale), else on all kernel headers
    if meta.get("kernels_main_sha16"):
        had, now = meta["kernels_main_sha16"], kernels_sha16(KERNEL_SOURCES_MAIN)
    else:
        had, now = meta.get("kernels_sha16", "?"), kernels_sha16()
    src = "profiles/%s (%d particles, 2 x FETCH_SIZE + WRITE_SIZE, KB; kernel sources %s%s)" % (
        summary, meta["particles"], had, " = the timed ones" if had == now else ", the timed ones are " + now)
    return per_particle * particles_per_launch, src


def pmc_traffic_sum(summary, kernels):
    """HBM-side bytes per profiled unit summed over several kernels (2 x FETCH_SIZE + WRITE_SIZE, KB; all their dispatches)."""
    tot, meta_, seen = 0.0, None, []
    for kname in kernels:
        e, meta = pmc_entry(summary, kname, merge=True)
        if not e:                           # a kernel the profiled build did not launch (k_sva_gather16 since the z pass emits the samples)
            continue
        if "FETCH_SIZE" not in e or "WRITE_SIZE" not in e or not meta.get("particles"):
            return None, "no FETCH_SIZE / WRITE_SIZE summary for %s under profiles/%s" % (kname, summary)
        tot += (2.0 * e["FETCH_SIZE"]["sum"] + e["WRITE_SIZE"]["sum"]) * 1024.0 / meta["particles"]
        meta_ = meta; seen.append(kname)
    if not seen:
        return None, "none of %s under profiles/%s" % (" / ".join(kernels), summary)
    had, now = meta_.get("kernels_sha16", "?"), kernels_sha16()
    return tot, "profiles/%s (%d units, 2 x FETCH_SIZE + WRITE_SIZE, KB, summed over %s; kernel sources %s%s)" % (
        summary, meta_["particles"], " + ".join(seen), had, " = the timed ones" if had == now else ", the timed ones are " + now)


def pmc_valu(summary, kernel, slices_per_particle):
    """Vector-instruction counters of the same summary: wave-instructions per particle and per stored slice."""
    e, meta = pmc_entry(summary, kernel)
    if not e or "SQ_INSTS_VALU" not in e or not meta.get("particles"):
        return None
    per_particle = e["SQ_INSTS_VALU"]["sum"] / meta["particles"]
    out = {"source": "profiles/" + summary, "SQ_INSTS_VALU_per_particle": round(per_particle, 1),
           "SQ_INSTS_VALU_per_slice": round(per_particle / slices_per_particle, 1)}
    for c in ("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        if c in e:
            out[c + "_per_particle"] = round(e[c]["sum"] / meta["particles"], 1)
    return out


def global_flops_per_slice(band, search_range_px, box, mask_radius_px):
    """Executed fp32 operations of k_global per STORED slice (it serves psi and psi + 180): per lane and row PAIR
    16 R FMA-flops for the shift rows (4 R packed FMAs on 2-vectors), 20 for the even / odd (A, Bq) parts and 4 for the
    two row sums; about 800 per lane for the window reduction and the recombination.  (The model norm sum C2 |P|^2 is a
    separate fp32 MFMA product, k_slice_norms, reported as `norms_mfma`.)  Row pairs = HsP / 2 with
    HsP = 2 (Bs + 1) rounded up to the loop's trip (16 rows for R <= 3, 8 above); 64 lanes, masked lanes included
    (they execute).  DESIGN.md §4."""
    Bs = int(np.ceil(band)) - 1
    Ns = 2
    while Ns < 2 * (Bs + 1):
        Ns *= 2
    # the window in search-grid steps as ppm_geom.h derives it: range 0 = the mask radius, at most Ns / 2 - 1 steps; windows wider
    # than 6 steps either side go to the full-window transform (k_gfft); forced onto k_global (PPM_GLOBAL_PATH=tiles) they are searched as
    # ntiles overlapping tiles of 13 x 13 steps, one k_global launch each (ppm_refine_batch)
    rng_px = search_range_px if search_range_px > 0 else mask_radius_px
    RS = max(1, min(int(np.ceil(rng_px / (box / Ns))), Ns // 2 - 1))
    R = min(RS, 6)            # k_global's register window (wider windows go to k_gfft unless PPM_GLOBAL_PATH=tiles)
    tiles_1d = (2 * RS + 1 + 2 * R) // (2 * R + 1)
    trip = 16 if R <= 3 else 8
    HsP = ((2 * (Bs + 1) + trip - 1) // trip) * trip
    return (64 * (HsP // 2) * (16 * R + 20 + 4) + 64 * 800) * tiles_1d * tiles_1d, R, HsP, tiles_1d * tiles_1d


def gfft_model(band, search_range_px, box, mask_radius_px):
    """Executed fp32 operations of k_gfft (ppm_gfft.h) per STORED slice and particle, and the bytes its column pass pulls through the
    CU's vector L1.  Column pass, 4 L threads (two orientations x two row parities x L columns): per bank row one v_pk_mul + three
    v_pk_fma (14 flop), the decimation twiddles on half the threads (6 per row), one L-point transform; row pass, 2 (2 RSy + 1)
    threads: L/2 - 1 pairs of the half-length trick (18 flop), one L-point transform, a penalty add and a maximum per column (4 L).
    L-point transform: radix-4 butterflies of 8 packed adds (16 flop), 3 complex twiddle products (18 flop) on the twiddled ones,
    a last radix-2 pass (4 flop per butterfly) when log2 L is odd.  DESIGN.md 4c."""
    Bs = int(np.ceil(band)) - 1
    Ns = 2
    while Ns < 2 * (Bs + 1):
        Ns *= 2
    rng_px = search_range_px if search_range_px > 0 else mask_radius_px
    RS = max(1, min(int(np.ceil(rng_px / (box / Ns))), Ns // 2 - 1))
    L = Ns // 2
    lg = int(np.log2(L))
    fft = (lg // 2) * (L // 4) * 16
    M = L
    while M >= 8:
        fft += (L // M) * (M // 4 - 1) * 18
        M //= 4
    if lg % 2:
        fft += (L // 2) * 4
    col = 4 * L * (L * 14 + L * 3 + fft)
    row = 2 * (2 * RS + 1) * ((L // 2 - 1) * 18 + fft + 4 * L)
    return {"flops_per_slice": float(col + row), "Ns": Ns, "L": L, "RS": RS, "fft_flops": fft,
            "l1_bytes_per_slice": L * L * 16.0,     # the slice is staged through LDS once per block (a quarter per wave)
            "lds_bytes_per_slice": (1.0 + 4.0) * L * L * 16.0 + 4.0 * L * L * 16.0 + 2.0 * (2 * RS + 1) * (L + 2) * 8.0 * 2}   # staging write + four waves' reads, W reads, T written and read


def default_search_bench(a, ref, stack, start_rows, truth, vol, px, cpu):
    """PYP's own default call of the grid search: "search range X / Y" = 0 = the mask radius (frealign.py:3954-3957,
    config/pyp_config.toml:5338-5350).  At 256^2 / 4 A that is a window of +-41 steps of the 128-point search grid, 83 x 83 shifts per
    orientation, which goes to the full-window transform kernel k_gfft instead of k_global's register-held 7 x 7 window.  Two
    configurations on particles of the SAME resident stack: the headline's (15 deg, band r) and PYP's defaults (20 deg, search
    limit 10 A); wall clock around the library call with inputs resident, kernel times from the library's HIP events."""
    import torch
    from pyp_amd import host, synth
    from pyp_amd.abi import RefineCfg
    N, M = a.box, len(start_rows)
    res = px * N / a.band
    out = {"what": "search range X / Y = 0 = the mask radius, as PYP sends it (frealign.py:3954-3957, config/pyp_config.toml:5338-5350)"}
    for key, step, res_search, npart in (("mask_radius_window_15deg_r%g" % a.band, a.angular_step, res, min(M, 16384)),
                                         ("pyp_defaults_20deg_10A", 20.0, 10.0 * px, min(M, 16384))):
        cfg = RefineCfg.make(box=N, pixel_size=px, mask_radius=0.32 * N * px, res_high=res, res_search=res_search, res_low=0.0,
                             angular_step=step, top_hits=20, search_range_x=0.0, search_range_y=0.0, res_signed_cc=30.0, molecular_mass_kda=500.0)
        sub, rows0 = stack[:npart], start_rows[:npart]
        ref.refine(cfg, stack[:256], start_rows[:256])                 # bank, tables, code objects
        host.profile(True, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = ref.refine(cfg, sub, rows0)
        host.lib.load().ppm_device_sync()
        dt = time.perf_counter() - t0
        prof = host.profile_report()
        host.profile(False, False)
        counts = ref.last_counts()
        band_s = px * N / res_search
        mdl = gfft_model(band_s, 0.0, N, 0.32 * N)
        nsl = counts["n_global"] / 2.0
        launches = max(prof["global"]["launches"], 1)
        ms_g = prof["global"]["ms"] / launches
        ppl = npart / launches
        tf = ppl * nsl * mdl["flops_per_slice"] / (ms_g * 1e-3) / 1e12
        l1 = ppl * nsl * mdl["l1_bytes_per_slice"] / (ms_g * 1e-3) / 1e9
        gf_traffic, gf_src = (pmc_traffic(pmc_latest("refine0"), "k_gfft<%d" % int(np.log2(mdl["Ns"])), ppl) if key.startswith("mask_radius")
                              else (None, "counters are taken for the 256^2 / 15 deg / r = 64 case only (profiles/r05_pmc_refine0.json)"))
        k = min(npart, 2000)
        ang = synth.angular_error_deg(got[:k], truth[:k])
        blk = {"value": round(npart / dt, 1), "unit": "particles/s", "particles": npart, "wall_s": round(dt, 3),
               "config": {"box": N, "angular_step": step, "res_search_A": round(res_search, 3), "search_band_px": round(band_s, 2), "orientations": counts["n_global"],
                          "search_grid_points": mdl["Ns"], "window_steps_each_side": mdl["RS"], "shifts_per_orientation": (2 * mdl["RS"] + 1) ** 2},
               "kernels_us_per_particle": {k2: round(v["ms"] * 1e3 / npart, 3) for k2, v in prof.items() if v["launches"]},
               "roofline": {"bound": "valu_fp32", "kernel": "k_gfft", "achieved": round(tf, 2), "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_VALU_TFLOPS, 4),
                            "traffic": gf_traffic, "traffic_source": gf_src, "avg_launch_ms": round(ms_g, 3), "particles_per_launch": round(ppl, 1),
                            "flops_per_stored_slice_and_particle": mdl["flops_per_slice"],
                            "flop_model": "column pass 4 L x (17 L + T) + row pass 2 (2 RS + 1) x (9 L + T - 18 + 4 L) with T = %d flop per %d-point transform (gfft_model)" % (mdl["fft_flops"], mdl["L"]),
                            "l1_path": {"GBps": round(l1, 1), "peak": PEAK_L1_GBPS, "frac": round(l1 / PEAK_L1_GBPS, 4),
                                        "note": "bank rows pulled through the vector L1: the slice once per block (each wave stages a quarter into LDS; four times the slice before the staging); "
                                                "peak derived: 64 B/clk/CU x 256 CUs x 2.4 GHz"},
                            "lds_path": {"GBps": round(ppl * nsl * mdl["lds_bytes_per_slice"] / (ms_g * 1e-3) / 1e9, 1), "peak": round(256 * 128 * 2.4, 1),
                                         "frac": round(ppl * nsl * mdl["lds_bytes_per_slice"] / (ms_g * 1e-3) / 1e9 / (256 * 128 * 2.4), 4),
                                         "note": "LDS bytes of a pass (slice staged and read by four waves, W table reads, image T written and read); peak derived: 128 B/clk/CU x 256 CUs x 2.4 GHz"}},
               "accuracy_vs_truth": {"median_deg": round(float(np.median(ang)), 3), "frac_within_2deg": round(float((ang < 2).mean()), 3)}}
        if cpu and key.startswith("mask_radius"):
            # the same particles through the oracle's zero-filled inverse transform (ccf_mode 0), all host cores
            import ctypes
            from oracle import oracle
            cores = host_cores()
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
            n = min(npart, cores)
            oref = oracle.Reference(vol, N / 2)
            t0 = time.time()
            want, _ = oracle.refine_batch(oref, cfg, sub[:n].cpu().numpy(), rows0[:n], ccf_mode=0)
            tc = time.time() - t0
            oref.close()
            blk["parity_vs_oracle"] = pose_parity(want, got[:n], px, "the first %d particles of this run against the oracle's zero-filled inverse transform (ccf_mode 0)" % n)
            blk["cpu_baseline"] = {"value": round(n / tc, 3), "unit": "particles/s", "cores": cores, "kind": "port",
                                   "sample": "%d particles, %.1f s wall, OpenMP over particles, oracle ccf_mode 0 (zero-filled %d^2 inverse transform per orientation)" % (n, tc, mdl["Ns"])}
        out[key] = blk
    return out


# --------------------------------------------------------------------------------------------- main
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a, argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1) and rank == 0:
        print(f"WARNING: --gpus {a.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    if os.environ.get("PPM_BENCH_PROBE") == "1":
        return probe_main(a, rank, world)
    if "PPM_FORCE_DEVICE" in os.environ:          # rehearsal of the N > 1 path on a one-GPU box (with PPM_DIST_BACKEND=gloo)
        local = int(os.environ["PPM_FORCE_DEVICE"])
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("ERROR: bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
        return 2
    if local >= torch.cuda.device_count():
        print(f"ERROR: rank {rank} wants device {local}, only {torch.cuda.device_count()} visible", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    if world > 1:
        backend = os.environ.get("PPM_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    ctx = dict(a=a, rank=rank, world=world, local=local, dev=torch.device("cuda", local), px=1.0)
    from pyp_amd import synth
    ctx["vol"] = synth.phantom(a.box)
    line = None
    if a.workload in ("both", "refine"):
        line = refine_bench(ctx)
    if a.workload in ("both", "reconstruct"):
        rec = reconstruct_bench(ctx)
        if rank == 0:
            if line is None:
                line = rec
            else:
                line["reconstruct"] = rec
    for name, fn in (("csp", csp_bench), ("sva", sva_bench)):
        if a.workload == name or (a.workload == "both" and not a.no_next_rows):
            blk = fn(ctx)
            if rank == 0:
                if line is None:
                    line = blk
                else:
                    line[name] = blk
    if rank == 0:
        line["summary"] = summary_of(line)          # LAST key: the figures a reader of the line's tail needs
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def summary_of(line):
    """The handful of figures that decide the reading of the line, gathered under its LAST key (a record that keeps only the tail of
    the 15 KB line still shows them): headline value and parity, PYP's default search, reconstruction value and parity, the
    executables, the next rows, and what the collective ran on."""
    def pick(d, *keys):
        for k in keys:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d

    def par(d):
        return None if not isinstance(d, dict) else {k: d[k] for k in ("n", "max_deg", "max_shift_px", "rel_l2_accumulator", "see", "error") if k in d}
    main_is_refine = "particles/sec projection-matching" in str(line.get("metric", ""))
    rec = line if not main_is_refine and "Fourier insertion" in str(line.get("metric", "")) else line.get("reconstruct")
    s = {"metric": line.get("metric"), "value": line.get("value"), "unit": line.get("unit"), "n_gpus": line.get("n_gpus"), "ms_per_step": line.get("ms_per_step"),
         "roofline_frac": pick(line, "roofline", "frac"), "parity_vs_oracle": par(line.get("parity_vs_oracle")), "cpu_baseline_value": pick(line, "cpu_baseline", "value")}
    ds = line.get("default_search")
    if isinstance(ds, dict):
        s["default_search"] = {k: {"value": v.get("value"), "roofline_frac": pick(v, "roofline", "frac"), "parity_vs_oracle": par(v.get("parity_vs_oracle"))}
                               for k, v in ds.items() if isinstance(v, dict)} or {"error": ds.get("error")}
    if isinstance(rec, dict):
        s["reconstruct"] = {"value": rec.get("value"), "unit": rec.get("unit"), "roofline_frac": pick(rec, "roofline", "frac"),
                            "parity_vs_oracle": par(rec.get("parity_vs_oracle")), "collective": rec.get("collective")}
    dr = line.get("dropin")
    if isinstance(dr, dict):
        s["dropin"] = {"refine3d": pick(dr, "refine3d", "value"), "refine3d_search_range_0": pick(dr, "refine3d_search_range_0", "value"),
                       "reconstruct3d": pick(dr, "reconstruct3d", "value"),
                       "resident_server": {k: pick(dr, "resident_server", k, "value") for k in ("refine3d", "reconstruct3d", "refine3d_again")} if "resident_server" in dr else None,
                       "error": dr.get("error")}
    for name in ("csp", "sva"):
        b = line.get(name)
        if isinstance(b, dict):
            s[name] = {"value": b.get("value"), "unit": b.get("unit"), "roofline_frac": pick(b, "roofline", "frac"),
                       "parity_vs_oracle": par(b.get("parity_vs_oracle")) or pick(b, "parity_vs_oracle", "max_deg")}
    return s


def make_barrier(world):
    import torch
    import torch.distributed as dist
    from pyp_amd import host

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        host.lib.load().ppm_device_sync()
    return barrier


def max_over_ranks(dt, world, dev):
    import torch
    import torch.distributed as dist
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def refine_bench(ctx):
    import torch
    from pyp_amd import host, synth
    from pyp_amd.abi import RefineCfg
    a, rank, world, local, dev, px, vol = (ctx[k] for k in ("a", "rank", "world", "local", "dev", "px", "vol"))
    N, M = a.box, a.particles
    # ---- synthetic inputs: phantom, one pose per particle, CTF, SNR 0.05; rank r gets its own poses / noise
    uniq = M if a.unique <= 0 else min(a.unique, M)
    _, stack, rows = synth.make_dataset(N, M, pixel=px, snr=0.05, vol=vol, device=dev, unique=uniq,
                                        seed_poses=synth.SEED_POSES + rank, seed_noise=synth.SEED_NOISE + rank, batch=32)
    torch.cuda.synchronize()
    res = px * N / a.band
    srange = a.search_range * px
    cfg = RefineCfg.make(box=N, pixel_size=px, mask_radius=0.32 * N * px, res_high=res, res_search=res, res_low=0.0,
                         angular_step=a.angular_step, top_hits=20, search_range_x=srange, search_range_y=srange,
                         res_signed_cc=30.0, molecular_mass_kda=500.0)
    t0 = time.time()
    ref = host.Reference(vol, N / 2, device=local)
    t_refprep = time.time() - t0
    start_rows = synth.cistem.default_rows(M, px, 300.0, 2.7, 0.07)      # from-scratch rows: the search ignores the poses
    for c in ("DEFOCUS_1", "DEFOCUS_2", "DEFOCUS_ANGLE"):
        start_rows[:, synth.cistem.COL[c]] = rows[:, synth.cistem.COL[c]]
    barrier = make_barrier(world)
    out = None
    for _ in range(a.warmup):
        out = ref.refine(cfg, stack, start_rows)
    host.profile(True, True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = ref.refine(cfg, stack, start_rows)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    prof = host.profile_report()
    host.profile(False, False)
    counts = ref.last_counts()
    ref.close()
    if rank != 0:
        del stack
        torch.cuda.empty_cache()
        return None
    total = world * M * a.steps
    value = total / dt
    # ---- roofline of the dominant kernel, from the launch durations the library measured with HIP events on ITS stream
    # around every launch of the timed region
    S_g = counts["samples_global"]
    launches_g = max(prof["global"]["launches"], 1)
    ppl = M * a.steps / launches_g                                            # particles per k_global launch (a tiled shift window launches once per tile: see below)
    ms_g = prof["global"]["ms"] / launches_g
    fl_slice, R, HsP, ntiles = global_flops_per_slice(a.band, srange / px, N, 0.32 * N)
    gm_ = gfft_model(a.band, srange / px, N, 0.32 * N)
    use_gfft = gm_["RS"] > 6 and gm_["Ns"] >= 16 and os.environ.get("PPM_GLOBAL_PATH") != "tiles"       # the library's own rule (ppm_refine_batch)
    if use_gfft:
        fl_slice, ntiles = gm_["flops_per_slice"], 1
    n_slices = counts["n_global"] / 2.0                                        # stored slices (psi and psi + 180 share one)
    flops_g = ppl * n_slices * fl_slice
    tf = flops_g / (ms_g * 1e-3) / 1e12
    bytes_model = ppl * counts["n_global"] * 8.0 * S_g * ntiles               # SURVEY §8(d) streaming model (every tile of the shift window streams the bank again)
    gbps_model = bytes_model / (ms_g * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(pmc_latest("refine"), "k_global", ppl)
    roof = {"bound": "valu_fp32", "kernel": "k_gfft" if use_gfft else "k_global", "achieved": round(tf, 2), "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_VALU_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(ms_g, 3),
            "particles_per_launch": round(ppl * ntiles, 1), "shift_window_tiles": ntiles, "flops_per_launch": flops_g,
            "flop_model": ("stored slices (%d) x gfft_model(): full-window transform, %d-point search grid, window +-%d steps (DESIGN.md 4c)" % (int(n_slices), gm_["Ns"], gm_["RS"])) if use_gfft else
                          ("stored slices (%d) x [64 lanes x %d row pairs x (16 R + 24) + 64 x 800], R = %d shift rows; executed fp32 "
                           "operations incl. masked lanes (DESIGN.md §4)%s" % (int(n_slices), HsP // 2, R, "" if ntiles == 1 else " x %d tiles of the shift window" % ntiles)),
            "in_band_fraction_of_lane_rows": round(S_g / (64.0 * HsP), 3),
            "hbm_streaming_model": {"bytes_per_launch": bytes_model, "GBps": round(gbps_model, 1),
                                    "frac_clamped": round(min(1.0, gbps_model / PEAK_HBM_GBPS), 4), "exceeds_peak": bool(gbps_model > PEAK_HBM_GBPS),
                                    "note": "SURVEY §8(d) contract figure 8 S(r) bytes per orientation; the 144 MB slice bank is shared by all "
                                            "particles and served from L2 / Infinity Cache, so this is not HBM traffic (see traffic)"},
            "hbm_traffic_frac": None if traffic is None else round(traffic / (ms_g * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)}
    if prof.get("norms", {}).get("launches"):
        # the slice norms: n x n_slices x K fp32 MFMA product per launch, K = (2 Bs + 1) bank rows x 64 columns
        Bs_ = int(np.ceil(a.band)) - 1
        ms_n = prof["norms"]["ms"] / prof["norms"]["launches"]
        fl_n = 2.0 * ppl * n_slices * (2 * Bs_ + 1) * 64
        roof["norms_mfma"] = {"kernel": "k_slice_norms", "avg_launch_ms": round(ms_n, 3), "flops_per_launch": fl_n,
                              "achieved": round(fl_n / (ms_n * 1e-3) / 1e12, 2), "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(fl_n / (ms_n * 1e-3) / 1e12 / PEAK_VALU_TFLOPS, 4),
                              "note": "v_mfma_f32_32x32x2_f32: the fp32 matrix peak equals the fp32 vector peak (256 flop / cycle / CU)"}
    roof["in_band_only_frac"] = round(tf / PEAK_VALU_TFLOPS * S_g / (64.0 * HsP), 4)      # `frac` counts masked lanes (they execute); this is the share of the peak spent on in-band samples
    # k_local (second kernel of the step): one trilinear gather of the reference per (sample, rotation) - 4 x 16-byte loads served by
    # L2 / L1 - and ~70 fp32 operations on it (position 12, seven complex interpolations 42, CTF 2, |m|^2 3, phase 6, dot 4, ring sums 1)
    if prof.get("local", {}).get("launches"):
        ms_l = prof["local"]["ms"] / (M * a.steps) * 1e-3                                  # seconds per particle
        gathers = counts["samples_local"]                                                   # gathered samples per particle (every evaluation at its marching band)
        fl_l = 70.0 * gathers
        gbps_l = 64.0 * gathers / ms_l / 1e9
        roof["local"] = {"kernel": "k_local", "bound": "l1_gather", "us_per_particle": round(ms_l * 1e6, 3), "gathered_samples_per_particle": gathers,
                         "achieved": round(gbps_l, 1), "peak": PEAK_L1_GBPS, "unit": "GB/s", "frac": round(gbps_l / PEAK_L1_GBPS, 4),
                         "bytes_model": "64 B per gathered sample: the 2 x 2 x 2 neighbourhood of the reference as four 16-byte loads through the CU's vector L1",
                         "peak_note": PEAK_L1_NOTE,
                         "useful_TFLOPs": round(fl_l / ms_l / 1e12, 2),
                         "flop_model": "70 fp32 operations per gathered sample (useful arithmetic; the kernel issues ~75 vector instructions per gathered sample)",
                         "note": "bound by the vector memory path (DESIGN.md 4b): 64 cycles per 64-lane gather at 64 B/clock/CU plus ~6 line fills from L2; "
                                 "probes with all L1 hits / without LDS atomics / with half the instructions are in CHANGELOG.md round 4"}
    pv = pmc_valu(pmc_latest("refine"), "k_global", n_slices)
    if pv:
        pv["model_flops_per_lane_instruction"] = round(fl_slice / (pv["SQ_INSTS_VALU_per_slice"] * 64.0), 3)
        roof["pmc"] = pv
    b_pm = 4.0 * N * N + counts["n_global"] * 8.0 * S_g + 8.0 * counts["samples_local"] + 128
    per_particle_us = {k2: round(v["ms"] * 1e3 / (M * a.steps), 3) for k2, v in prof.items() if v["launches"]}
    k = min(M, 2000)
    ang = synth.angular_error_deg(out[:k], rows[:k])
    shf = synth.shift_error_px(out[:k], rows[:k], px)
    line = {
        "metric": "particles/sec projection-matching, 256^2 box", "value": round(value, 1), "unit": "particles/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "SPA global search: %dk %d^2 particles/GPU, %g deg angular step, band r=%g px, top-20 hits refined + best hit "
                               "continued at the full band" % (M // 1000, N, a.angular_step, a.band),
                   "particles_per_gpu": M, "distinct_poses_per_gpu": uniq, "box": N, "orientations": counts["n_global"],
                   "local_evaluations": counts["n_local"], "parallelism": "particle-sharded x%d" % world},
        "roofline": roof,
        "compulsory_bytes_per_particle": 4 * N * N + 128, "contract_bytes_per_particle": b_pm,
        "kernels_ms": {k2: round(v["ms"], 2) for k2, v in prof.items() if v["launches"]},
        "kernels_us_per_particle": per_particle_us,
        "reference_prep_s": round(t_refprep, 3), "library": so_sha16(), "kernel_sources": kernels_sha16(),
        "accuracy_vs_truth": {"median_deg": round(float(np.median(ang)), 3), "frac_within_2deg": round(float((ang < 2).mean()), 3),
                              "median_shift_px": round(float(np.median(shf)), 3)},
    }
    if world == 1 and not a.no_cpu and a.cpu_seconds > 0:
        line["cpu_baseline"], line["parity_vs_oracle"] = cpu_baseline(vol, stack, start_rows, cfg, N, a.cpu_seconds, out, px)
    elif world > 1:
        line["cpu_baseline"] = {"see": "N=1 line (the CPU legs run on rank 0 of the one-GPU run only)"}
        line["parity_vs_oracle"] = {"see": "N=1 line"}
    if world == 1 and not a.no_side:
        try:                    # side figures: a failure is reported in the line
            ref2 = host.Reference(vol, N / 2, device=local)
            line["default_search"] = default_search_bench(a, ref2, stack, start_rows, rows, vol, px, (not a.no_cpu) and a.cpu_seconds > 0)
            ref2.close()
        except Exception as e:          # noqa: BLE001
            line["default_search"] = {"error": str(e)[:300]}
    if world == 1 and not a.no_dropin:
        try:                    # side figures: a failure is reported in the line
            line["dropin"] = dropin_bench(a, vol, stack, start_rows, rows, px, res, srange)
        except Exception as e:          # noqa: BLE001
            line["dropin"] = {"error": str(e)[:300]}
    del stack
    torch.cuda.empty_cache()
    return line


def dropin_bench(a, vol, stack, start_rows, truth, px, res, srange):
    """The boundary PYP really calls: `bin/refine3d` and `bin/reconstruct3d` as child processes, answers on stdin in the order of
    frealign.py:3918-3994 / :1780-1824 (PYP's default flags: global = yes, local = no, 20 hits; C1; the shift search range of the
    headline configuration, --search-range, and once more with PYP's default range 0 = the mask radius), the SAME stack as a file
    (memory-backed, so the page cache is warm like a node-local scratch copy), one process for the whole range.  Wall time of the
    child from start to exit: interpreter, GPU context, reference preparation, reading, uploads, outputs - all included."""
    import shutil
    import tempfile
    from pyp_amd.formats import cistem, mrc
    M, N = len(start_rows), a.box
    need = M * N * N * 4 + (1 << 30)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    if shutil.disk_usage(base).free < need:
        return {"skipped": "%s has less than %.0f GB free for the stack file" % (base, need / 1e9)}
    d = tempfile.mkdtemp(prefix="ppm_dropin_", dir=base)
    try:
        t0 = time.time()
        mm = mrc.create(os.path.join(d, "p_stack.mrc"), (M, N, N), pixel_size=px)
        step = max(1, (1 << 30) // (N * N * 4))
        for lo in range(0, M, step):
            mm[lo:lo + step] = stack[lo:lo + step].cpu().numpy()
        mm.flush()
        del mm
        mrc.write(vol, os.path.join(d, "p_r01.mrc"), pixel_size=px)
        cistem.write_parameters(os.path.join(d, "p_r01.cistem"), start_rows)
        t_write = time.time() - t0
        # one read pass before the timings: the FIRST pass over a freshly written tmpfs file runs at a fifth of the page-cache rate on
        # this host (12 vs 55 GB/s, scripts/shm_read_probe.py) - an artefact of the file's age, not of the executables; PYP's stacks
        # are written once and read by every later program
        from concurrent.futures import ThreadPoolExecutor
        fd = os.open(os.path.join(d, "p_stack.mrc"), os.O_RDONLY)
        size = os.path.getsize(os.path.join(d, "p_stack.mrc"))

        def _warm(t, parts=8):
            buf = bytearray(64 << 20)
            pos, end = t * (size // parts), size if t == parts - 1 else (t + 1) * (size // parts)
            while pos < end:
                got = os.preadv(fd, [memoryview(buf)[:min(len(buf), end - pos)]], pos)
                if got <= 0:
                    break
                pos += got
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(_warm, range(8)))
        os.close(fd)
        rng = "%07d_%07d" % (1, M)
        refine = ["p_stack.mrc", "p_r01.cistem", "null", "p_r01.mrc", "statistics_r01.txt", "no", "no", f"p_r01_match.mrc_{rng}", f"p_r01_{rng}.cistem",
                  f"p_r01_{rng}_changes.cistem", "C1", 1, M, 1, px, 500.0, 0, 0.32 * N * px, 0.0, res, 30.0, 8.0, 1.5 * 0.32 * N * px, res, a.angular_step, 20,
                  srange, srange, 0, 0, 0, 0, 500, 50.0, 1, "yes", "no", "yes", "yes", "yes", "yes", "yes", "no", "no", "no", "yes", "no", "no", "no", "no"]
        recon = ["p_stack.mrc", f"p_r01_{rng}.cistem", "null", "p_r01.mrc", "p_r01_map1.mrc", "p_r01_map2.mrc", "output.mrc", "p_r01_n1.res", "C1", 1, M, px, 500.0,
                 0, 0.45 * N * px, 2 * px, 0, 2.0, "no", 0, -1, "no", 0, 1, 1, "yes", "yes", "no", "no", "no", "yes", "no", "no", "no", "no", "yes",
                 os.path.join(d, "p_r01_map1_n1.mrc"), os.path.join(d, "p_r01_map2_n1.mrc"), 1]
        out = {"stack_file": "%d x %d^2 float32 = %.1f GB in %s (written in %.1f s, outside the timings)" % (M, N, M * N * N * 4 / 1e9, base, t_write)}
        for prog, script, sentinel in (("refine3d", refine, "Refine3D: Normal termination"), ("reconstruct3d", recon, "Reconstruct3D: Normal termination")):
            cmd = f"{ROOT}/bin/{prog} << eot > {prog}.log 2>&1\n" + "\n".join(str(x) for x in script) + "\neot\n"
            runs = []               # three runs, the median one reported (the host's page-cache read rate varies run to run)
            for _ in range(3):
                t0 = time.time()
                rc = subprocess.run(cmd, shell=True, cwd=d).returncode
                dt = time.time() - t0
                log = open(os.path.join(d, prog + ".log")).read()
                if rc != 0 or sentinel not in log:
                    out[prog] = {"error": log[-600:]}
                    break
                timing = [ln for ln in log.splitlines() if ln.startswith("Timing:")]
                pipe = [ln for ln in log.splitlines() if ln.startswith("Pipeline:")]
                runs.append((dt, timing[0][8:] if timing else None, pipe[0][10:] if pipe else None))
            if "error" in out.get(prog, {}):
                break
            dt, timing, pipe = sorted(runs)[1]
            out[prog] = {"value": round(M / dt, 1), "unit": "particles/s", "wall_s": round(dt, 2), "wall_s_all_runs": [round(r[0], 2) for r in runs],
                         "phases": timing, "pipeline": pipe}
        # ---- refine3d exactly as PYP's default parameters call it: answers 27 / 28 = 0 = the mask radius (frealign.py:3954-3957)
        if "value" in out.get("refine3d", {}):
            refine0 = list(refine)
            refine0[26] = refine0[27] = 0
            cmd = f"{ROOT}/bin/refine3d << eot > refine3d_s0.log 2>&1\n" + "\n".join(str(x) for x in refine0) + "\neot\n"
            t0 = time.time()
            rc = subprocess.run(cmd, shell=True, cwd=d).returncode
            dt = time.time() - t0
            log = open(os.path.join(d, "refine3d_s0.log")).read()
            if rc != 0 or "Refine3D: Normal termination" not in log:
                out["refine3d_search_range_0"] = {"error": log[-400:]}
            else:
                timing = [ln for ln in log.splitlines() if ln.startswith("Timing:")]
                out["refine3d_search_range_0"] = {"value": round(M / dt, 1), "unit": "particles/s", "wall_s": round(dt, 2), "phases": timing[0][8:] if timing else None,
                                                  "note": "answers 27 / 28 = 0 = the mask radius: +-41 search-grid steps at 256^2 / 4 A (k_gfft); one run"}
        # ---- the same iteration with the resident per-GPU server (PPM_STACK_CACHE=1, pyp_amd/csrc/dropin_server.h): the first call starts
        # it and uploads the range, the calls after it find context, stack and reference in place
        if "value" in out.get("refine3d", {}) and "value" in out.get("reconstruct3d", {}):
            env = dict(os.environ, PPM_STACK_CACHE="1", PPM_LOCK_DIR=d, PPM_STACK_CACHE_IDLE_S="300")
            cached = {}
            try:
                seq = (("first_call_reconstruct3d", "reconstruct3d", recon), ("refine3d", "refine3d", refine), ("reconstruct3d", "reconstruct3d", recon),
                       ("refine3d_again", "refine3d", refine))
                for key, prog, script in seq:
                    cmd = f"{ROOT}/bin/{prog} << eot > {prog}_srv.log 2>&1\n" + "\n".join(str(x) for x in script) + "\neot\n"
                    t0 = time.time()
                    rc = subprocess.run(cmd, shell=True, cwd=d, env=env).returncode
                    dt = time.time() - t0
                    log = open(os.path.join(d, prog + "_srv.log")).read()
                    if rc != 0 or "Normal termination" not in log:
                        cached[key] = {"error": log[-400:]}
                        break
                    timing = [ln for ln in log.splitlines() if ln.startswith("Timing:")]
                    cached[key] = {"value": round(M / dt, 1), "unit": "particles/s", "wall_s": round(dt, 3), "phases": timing[0][8:] if timing else None,
                                   "stack_resident": "are resident in device memory" in log}
                cached["note"] = ("reconstruct3d -> refine3d -> reconstruct3d -> refine3d of one stack through bin/ppm_server: the first call creates the context and "
                                  "uploads the 26 GB once; the later ones are clients of the resident process (no context, no PCIe pass, reference kept)")
            finally:
                subprocess.run([f"{ROOT}/bin/ppm_server", "--stop"], env=env, capture_output=True, timeout=120)
            out["resident_server"] = cached
        if "value" in out.get("refine3d", {}):
            got = cistem.read_parameters(os.path.join(d, f"p_r01_{rng}.cistem"))
            k = min(M, 2000)
            out["refine3d"]["median_deg_vs_truth"] = round(float(np.median(__import__("pyp_amd.synth", fromlist=["x"]).angular_error_deg(got[:k], truth[:k]))), 3)
        out["note"] = "one child process per call, start-up included; 262 KB per particle over PCIe bounds any file-fed path at ~190 k particles/s"
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def reconstruct_bench(ctx):
    """configs[2]: every rank inserts its particles into private accumulators (a torch tensor handed to the
    library), then ONE all-reduce (sum, f32) over RCCL; finalisation on rank 0 is outside the timed region."""
    import torch
    from pyp_amd import dist as pdist
    from pyp_amd import host, synth
    from pyp_amd.abi import FinalCfg, ReconCfg
    a, rank, world, local, dev, px, vol = (ctx[k] for k in ("a", "rank", "world", "local", "dev", "px", "vol"))
    N, M = a.box, a.recon_particles
    uniq = min(M, 4096 if a.unique <= 0 else a.unique)     # distinct poses; every particle still has its own noise (insertion cost is pose-independent)
    _, stack, rows = synth.make_dataset(N, M, pixel=px, snr=0.05, vol=vol, device=dev, unique=uniq,
                                        seed_poses=synth.SEED_POSES + 100 + rank, seed_noise=synth.SEED_NOISE + 100 + rank, batch=32)
    torch.cuda.synchronize()
    nfl = int(host.lib.load().ppm_accum_floats(N))
    host.lib.init(local)
    acc_t = torch.zeros(nfl, dtype=torch.float32, device=dev)
    acc = host.Accumulator(N, px, "C1", device=local, ext_tensor=acc_t)
    rows = rows.copy()
    rows[:, 0] = np.arange(1, M + 1) + rank * M     # global positions: half assignment must not depend on the rank count
    rc = ReconCfg(box=N, pixel_size=px, res_limit=2 * px, score_weight_bfactor=0.0, score_average=0.0, score_threshold=0.0,
                  normalize=1, invert=0, split_by_pind=0, mask_radius=0.32 * N * px)
    barrier = make_barrier(world)

    # the one collective: through the C ABI (ppm_accum_reduce, the library's own RCCL communicator) when the ranks have a GPU
    # each; PPM_REDUCE=torch (or a gloo rehearsal with shared devices) sums the same tensor with torch.distributed instead
    import torch.distributed as tdist
    via_abi = world > 1 and tdist.get_backend() == "nccl" and os.environ.get("PPM_REDUCE", "abi") == "abi"

    def step():
        acc_t.zero_()
        acc.set_counts(0, 0)
        acc.insert(rc, stack, rows)
        if via_abi:
            return pdist.reduce_accumulator_handle(acc, local)
        return pdist.reduce_accumulators(acc_t, acc.counts())[1]

    abi_note = None
    if via_abi:
        # the communicator is made once, outside the timed region (collective).  RCCL has never seen more than one rank of this
        # library on the development pool (one GPU per box): if creating the communicator fails on any rank, every rank falls back
        # to torch.distributed's all-reduce of the same tensor and the line says so
        ok = 1
        try:
            pdist.library_comm(local)
        except Exception as e:          # noqa: BLE001
            ok, abi_note = 0, "ppm_comm_create failed (%s): torch.distributed all_reduce used instead" % str(e)[:200]
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            via_abi = False
            abi_note = abi_note or "ppm_comm_create failed on another rank: torch.distributed all_reduce used instead"
    # what the collective ran on, for the record: the communicator's size as RCCL itself reports it (ppm_comm_count = ncclCommCount)
    collective = {"library": None, "ranks": 1, "via_abi": False, "world_size": world}
    if world > 1:
        collective = {"library": "RCCL through ppm_accum_reduce (C ABI)" if via_abi else "torch.distributed all_reduce (%s)" % tdist.get_backend(),
                      "ranks": host.comm_count(pdist.library_comm(local)) if via_abi else tdist.get_world_size(), "via_abi": bool(via_abi), "world_size": world}
        if abi_note:
            collective["note"] = abi_note
    counts = None
    for _ in range(a.warmup):
        counts = step()
    host.profile(True, True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        counts = step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    prof = host.profile_report()
    host.profile(False, False)
    if rank != 0:
        acc.close()
        return None
    S = int(np.floor(np.pi * (N / 2) ** 2 / 2))
    b_ins = 4.0 * N * N + S * 192.0
    per_us = {k2: round(v["ms"] * 1e3 / (M * a.steps), 3) for k2, v in prof.items() if v["launches"]}
    # dominant kernel of this workload: pre-processing (image -> band spectrum) or brick insertion
    dom = "prep" if prof["prep"]["ms"] >= prof["insert"]["ms"] else "insert"
    nl = max(prof[dom]["launches"], 1)
    ms = prof[dom]["ms"] / nl
    ppl = M * a.steps / nl
    kname = "k_prep" if dom == "prep" else "k_insert_bricks"
    traffic, traffic_src = pmc_traffic(pmc_latest("reconstruct"), kname, ppl)
    if dom == "prep":
        B = N // 2 - 1
        alg = ppl * (4.0 * N * N + 8.0 * (2 * B + 1) * (B + 1))       # image read once + band spectrum written once
        what = "4 N^2 image bytes read + 8 (2B+1)(B+1) band-spectrum bytes written per particle"
    else:
        B = N // 2 - 1
        alg = ppl * 8.0 * (2 * B + 1) * (B + 1)                        # band spectrum read once; the bricks live in LDS
        what = "band spectrum read once (8 (2B+1)(B+1) bytes per particle); accumulator bricks are written once per launch"
    gbps = alg / (ms * 1e-3) / 1e9
    model_gbps = ppl * S * 192.0 / (ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "kernel": kname, "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(ms, 3), "particles_per_launch": round(ppl, 1),
            "algorithmic_bytes": what,
            "scatter_model": {"bytes_per_particle": b_ins, "GBps": round(model_gbps, 1), "frac_clamped": round(min(1.0, model_gbps / PEAK_HBM_GBPS), 4),
                              "exceeds_peak": bool(model_gbps > PEAK_HBM_GBPS),
                              "note": "SURVEY §8(d) contract figure (what a scatter into HBM would move: S x 8 taps x 12 B x 2); the bricks are "
                                      "accumulated in LDS, so this is not HBM traffic"}}
    if prof.get("insert", {}).get("launches"):
        # what k_insert_bricks really runs on: LDS integer atomics.  Every in-band sample adds its 8 trilinear taps x (re, im, weight) = 24
        # 64-bit adds into the brick in LDS; a 64-lane ds_add_u64 occupies the LDS for ~7.7 cycles (scripts/micro/lds_atomic_bench)
        ms_i = prof["insert"]["ms"] / (M * a.steps) * 1e-3                  # seconds per particle in the kernel
        lane_atomics = 24.0 * S
        peak_at = 256 * 64.0 / 7.7 * 2.4e9
        roof["lds_atomic"] = {"kernel": "k_insert_bricks", "bound": "lds_atomic", "achieved": round(lane_atomics / ms_i / 1e9, 1), "peak": round(peak_at / 1e9, 1),
                              "unit": "G lane-atomics/s", "frac": round(lane_atomics / ms_i / peak_at, 4), "us_per_particle": round(ms_i * 1e6, 3),
                              "model": "24 ds_add_u64 per in-band sample (8 taps x re, im, weight), S = %d samples per particle" % S,
                              "peak_note": "derived: one 64-lane ds_add_u64 per 7.7 cycles and CU (measured, scripts/micro/lds_atomic_bench) x 256 CUs x 2.4 GHz; the "
                                           "kernel's vector instructions run beside them, so neither pipe is the whole bound"}
        e_pmc, m_pmc = pmc_entry(pmc_latest("reconstruct"), "k_insert_bricks")
        if e_pmc and m_pmc.get("particles") and "SQ_INSTS_VALU" in e_pmc:
            roof["lds_atomic"]["counters"] = {"source": "profiles/" + pmc_latest("reconstruct"),
                                              "valu_lane_instructions_per_sample": round(64.0 * e_pmc["SQ_INSTS_VALU"]["sum"] / m_pmc["particles"] / S, 1)}
            if "SQ_LDS_BANK_CONFLICT" in e_pmc and e_pmc.get("SQ_LDS_IDX_ACTIVE", {}).get("sum"):
                roof["lds_atomic"]["counters"]["lds_conflict_share"] = round(e_pmc["SQ_LDS_BANK_CONFLICT"]["sum"] / e_pmc["SQ_LDS_IDX_ACTIVE"]["sum"], 3)
    acc.set_counts(counts[0], counts[1])
    h1, h2, fl, stats = acc.finalize(FinalCfg(molecular_mass_kda=500.0, inner_radius=0.0, outer_radius=0.45 * N * px, mask_falloff=0.0))
    acc.close()
    cc = float(np.corrcoef(fl.ravel(), vol.ravel())[0, 1])
    extra = {}
    if world == 1 and not a.no_cpu and a.cpu_seconds > 0:
        try:                    # side figures: a failure is reported in the line
            extra["parity_vs_oracle"], extra["cpu_baseline"] = recon_parity(stack, rows, rc, N, px, local)
        except Exception as e:          # noqa: BLE001
            extra["parity_vs_oracle"] = {"error": str(e)[:300]}
    elif world > 1:
        extra["cpu_baseline"] = {"see": "N=1 line"}
        extra["parity_vs_oracle"] = {"see": "N=1 line"}
    return {**extra, "metric": "particles/sec Fourier insertion, 256^2 box", "value": round(world * M * a.steps / dt, 1), "unit": "particles/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "3D reconstruction: Fourier-insert %dk %d^2 particles/GPU (resident stack, %.0f GB) -> %d^3 half-maps, C1, "
                                   "one all-reduce" % (M // 1000, N, M * N * N * 4 / 1e9, N),
                       "particles_per_gpu": M, "box": N, "parallelism": "particle-sharded x%d" % world,
                       "collective": "none (one rank)" if world == 1 else ("ppm_accum_reduce (RCCL all-reduce, C ABI)" if via_abi else (abi_note or "torch.distributed all_reduce"))},
            "roofline": roof, "kernels_ms": {k2: round(v["ms"], 2) for k2, v in prof.items() if v["launches"]},
            "kernels_us_per_particle": per_us, "compulsory_bytes_per_particle": 4 * N * N,
            "path_hbm_frac_compulsory": round(world * M * a.steps * 4.0 * N * N / dt / 1e9 / PEAK_HBM_GBPS / world, 4),
            "map_cc_vs_truth": round(cc, 4), "fsc_at_half_nyquist": round(float(stats[N // 4 - 1, 3]), 4), "collective": collective}


def gather_roofline(kernel, summary_workload, ms_total, launches, gathers_total, units_total, unit_name, note):
    """Roofline entry of a gather-and-score kernel (k_sva_eval, k_csp_eval: the sweep of k_local over other pose tables): every
    gathered sample is one trilinear fetch of the reference (4 x 16-byte loads served by L1 / L2) and ~70 fp32 operations on it
    (position, seven complex interpolations, weights, phase, dot products - the k_local model of DESIGN.md 4a).  `achieved` = those
    operations per launch / the average launch duration from the library's HIP events; `traffic` = HBM-side bytes per launch from the
    committed --pmc summary of the same workload (2 x FETCH_SIZE + WRITE_SIZE, separate passes, scaled by units)."""
    launches = max(int(launches), 1)
    ms = ms_total / launches
    gpl, upl = gathers_total / launches, units_total / launches
    tf = 70.0 * gpl / (ms * 1e-3) / 1e12
    # HBM-side bytes of ONE launch: the summary's average per dispatch (every sweep is a dispatch over all profiled units), scaled by units
    e, meta = pmc_entry(pmc_latest(summary_workload), kernel, merge=True)
    traffic, src = None, "no FETCH_SIZE / WRITE_SIZE summary for %s under profiles/%s" % (kernel, pmc_latest(summary_workload))
    if e and "FETCH_SIZE" in e and "WRITE_SIZE" in e and meta.get("particles"):
        traffic = (2.0 * e["FETCH_SIZE"]["per_dispatch"] + e["WRITE_SIZE"]["per_dispatch"]) * 1024.0 * upl / meta["particles"]
        had, now = meta.get("kernels_sha16", "?"), kernels_sha16()
        src = "profiles/%s (%d units per dispatch, 2 x FETCH_SIZE + WRITE_SIZE, KB, per dispatch; kernel sources %s%s)" % (
            pmc_latest(summary_workload), meta["particles"], had, " = the timed ones" if had == now else ", the timed ones are " + now)
    pv = None
    if e and "SQ_INSTS_VALU" in e and meta.get("particles") and meta.get("gathers_per_unit"):
        pv = {"source": "profiles/" + pmc_latest(summary_workload),
              "SQ_INSTS_VALU_lane_instructions_per_gathered_sample": round(e["SQ_INSTS_VALU"]["sum"] * 64.0 / (meta["particles"] * meta["gathers_per_unit"]), 1)}
        for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "TCC_HIT_sum", "TCC_MISS_sum"):
            if c in e:
                pv[c] = e[c]["sum"]
    gbps = 64.0 * gpl / (ms * 1e-3) / 1e9
    return {"bound": "l1_gather", "kernel": kernel, "achieved": round(gbps, 1), "peak": PEAK_L1_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_L1_GBPS, 4),
            "bytes_model": "64 B per gathered sample (four 16-byte loads through the CU's vector L1; DESIGN.md 4b)", "peak_note": PEAK_L1_NOTE,
            "traffic": traffic, "traffic_source": src, "avg_launch_ms": round(ms, 4), "launches": launches, unit_name + "_per_launch": round(upl, 1),
            "gathered_samples_per_launch": round(gpl), "useful_TFLOPs": round(tf, 2),
            "flop_model": "70 fp32 operations per gathered sample (useful arithmetic)", "pmc": pv, "note": note}


def sva_eval_roofline(prof, lc, nv, steps, wedges):
    """k_sva_eval (+ k_sva_finish), ~70 % of an alignment step.  Gathered samples: band samples x gathered rotations per sub-volume summed
    over the sweeps (ppm_refine_last_counts), times the share of the band inside the missing wedge's complement (masked samples are skipped)."""
    fw = float(np.clip((np.asarray(wedges)[:, 1] - np.asarray(wedges)[:, 0]) / 180.0, 0, 1).mean())
    g = lc["samples_local"] * fw * nv * steps
    return gather_roofline("k_sva_eval", "sva", prof["local"]["ms"], prof["local"]["launches"], g, nv * steps * lc["n_local"], "states",
                           "device-resident compass: %d sweeps per call, every sweep one launch over all sub-volumes (x %d parts), no host wait between them; %.0f %% of the band lies inside the "
                           "tilt range; bound by the gathers' path through the vector L1 (DESIGN.md 9)" % (lc["n_local"], 4, 100 * fw))


def recon_parity(stack, rows, rc, N, px, local, want=2000, budget_s=30.0):
    """The first `want` particles of the timed stack inserted by the oracle (oracle.insert_batch, OpenMP) and by the HIP path into
    fresh accumulators: relative L2 distance of the two accumulators (value channels and weight channel), particle counts.
    The oracle runs in pieces of 250 particles and stops after `budget_s` seconds; both sides insert the same particles."""
    from oracle import oracle
    from pyp_amd import host
    cores = host_cores()
    _omp_threads(cores)
    acc_o, cnt_o = np.zeros(oracle.accum_floats(N), np.float32), np.zeros(2, np.int64)
    done, t0 = 0, time.time()
    while done < min(want, len(rows)):
        hi = min(done + 250, want, len(rows))
        oracle.insert_batch(acc_o, cnt_o, rc, "C1", stack[done:hi].cpu().numpy(), rows[done:hi])
        done = hi
        if time.time() - t0 > budget_s:
            break
    tc = time.time() - t0
    acc = host.Accumulator(N, px, "C1", device=local)
    acc.insert(rc, stack[:done], rows[:done])
    g, cnt_g = acc.download(), acc.counts()
    acc.close()
    go, gg = acc_o.reshape(-1, 3).astype(np.float64), g.reshape(-1, 3).astype(np.float64)
    rel = lambda x, y: float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300))
    par = {"n": int(done), "rel_l2_accumulator": float("%.3g" % rel(gg, go)), "rel_l2_values": float("%.3g" % rel(gg[:, :2], go[:, :2])),
           "rel_l2_weights": float("%.3g" % rel(gg[:, 2], go[:, 2])), "max_abs_diff_over_max": float("%.3g" % (np.abs(gg - go).max() / np.abs(go).max())),
           "counts_equal": bool(list(cnt_g) == [int(cnt_o[0]), int(cnt_o[1])]),
           "sample": "the first %d particles of the timed stack into fresh accumulators: HIP path against oracle.insert_batch" % done}
    cpu = {"value": round(done / tc, 2), "unit": "particles/s", "cores": cores, "kind": "port", "sample": "%d particles, %.1f s wall, OpenMP" % (done, tc)}
    return par, cpu


# --------------------------------------------------------------------------------------------- next rows
def _omp_threads(n):
    import ctypes
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))


def csp_pipeline(ctx, vol, stack, rows2, p2, parts_truth, tilts, cfg, cc, n, px, n_series=4):
    """BASELINE config 4 as the caller runs it per tilt series (src/pyp/align/core.py:958-1005, docs/cli/constrained.rst:12-15): `csp` mode -2
    crops the projections of every particle out of the 41 tilt images of 4096^2 (ppm_extract_boxes, one call per image, straight into a
    resident stack), the particle units are refined against the reference (ppm_csp_refine), and the series' projections are inserted
    into the rank's half-map accumulators at their refined poses (ppm_insert_batch); after the last series ONE reduce over the ranks.
    `n_series` series are in flight per GPU, each on its own reference handle and thread (config 4: 32 series on 8 GPUs); the insertions
    share the accumulator and are serialised by a lock.  The same synthetic series serves all of them (its images are resident)."""
    import threading
    import torch
    from pyp_amd import dist as pdist
    from pyp_amd import host, synth
    from pyp_amd.abi import FinalCfg, ReconCfg
    from pyp_amd.formats import cistem
    a, rank, world, local, dev = (ctx[k] for k in ("a", "rank", "world", "local", "dev"))
    C = cistem.COL
    nt, size = len(tilts), 4096
    order = np.lexsort((rows2[:, C["PIND"]], rows2[:, C["TIND"]]))          # tilt-major: the boxes of one image are consecutive in the stack
    rows_t = rows2[order]
    series, rows_p = synth.paste_tilt_series(stack[torch.as_tensor(order, device=dev)], rows_t, nt, shape=(size, size), seed=5 + rank)
    torch.cuda.synchronize()
    m = len(rows_p)
    tind = rows_p[:, C["IMIND"]].astype(int)
    off = np.searchsorted(tind, np.arange(nt + 1))
    coords = np.ascontiguousarray(rows_p[:, [C["ORIGINAL_X_POSITION"], C["ORIGINAL_Y_POSITION"]]])
    nfl = int(host.lib.load().ppm_accum_floats(n))
    acc_t = torch.zeros(nfl, dtype=torch.float32, device=dev)
    acc = host.Accumulator(n, px, "C1", device=local, ext_tensor=acc_t)
    rc = ReconCfg(box=n, pixel_size=px, res_limit=2 * px, score_weight_bfactor=0.0, score_average=0.0, score_threshold=0.0, normalize=1, invert=0,
                  split_by_pind=1, mask_radius=0.32 * n * px)
    refs = [host.Reference(vol, n / 2, device=local) for _ in range(n_series)]
    stacks = [torch.empty((m, n, n), dtype=torch.float32, device=dev) for _ in range(n_series)]
    lock, errs, results, stage = threading.Lock(), [], [None] * n_series, [[0.0, 0.0, 0.0] for _ in range(n_series)]
    torch.cuda.synchronize()

    def one_series(k):
        try:
            t0 = time.perf_counter()
            for t in range(nt):
                host.extract_boxes(series[t], coords[off[t]:off[t + 1]], n, 0.32 * n * px, px, out=stacks[k][off[t]:off[t + 1]], device=local)
            t1 = time.perf_counter()
            rr, pp, _ = refs[k].csp_refine(cfg, cc, stacks[k], rows_p, p2, tilts)
            t2 = time.perf_counter()
            with lock:
                acc.insert(rc, stacks[k], rr)
            t3 = time.perf_counter()
            results[k] = pp
            stage[k] = [t1 - t0, t2 - t1, t3 - t2]
        except Exception as e:          # noqa: BLE001 - reported in the line
            errs.append(str(e)[:300])
    one_series(0)                        # warm-up: code objects, work arrays of handle 0
    barrier = make_barrier(world)
    acc_t.zero_(); acc.set_counts(0, 0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        th = [threading.Thread(target=one_series, args=(k,)) for k in range(n_series)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        counts = pdist.reduce_accumulators(acc_t, acc.counts())[1] if world > 1 else acc.counts()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    for r in refs:
        r.close()
    if rank != 0:
        acc.close()
        return None
    if errs or any(r is None for r in results):
        acc.close()
        return {"error": errs[0] if errs else "a series returned nothing"}
    acc.set_counts(counts[0], counts[1])
    _, _, fl, _ = acc.finalize(FinalCfg(molecular_mass_kda=0.0, inner_radius=0.0, outer_radius=0.45 * n * px, mask_falloff=0.0))
    acc.close()

    def perr(x, y):
        return np.array([np.degrees(np.arccos(np.clip((np.trace(synth.euler_matrix(-u[4], -u[5], -u[6]).T @ synth.euler_matrix(-v[4], -v[5], -v[6])) - 1) / 2, -1, 1)))
                         for u, v in zip(x, y)])
    return {"value": round(world * n_series * m * a.steps / dt, 1), "unit": "projections/s", "series_per_s": round(world * n_series * a.steps / dt, 2),
            "ms_per_series": round(dt / (a.steps * n_series) * 1e3, 2), "series_in_flight_per_gpu": n_series,
            "stage_ms_one_series_in_flight": {"extract_41_images": round(np.mean([x[0] for x in stage]) * 1e3, 2), "refine": round(np.mean([x[1] for x in stage]) * 1e3, 2),
                                               "insert": round(np.mean([x[2] for x in stage]) * 1e3, 2),
                                               "note": "wall time per stage inside a worker while the other series run (they overlap)"},
            "tilt_series": "%d tilt images of %d^2 (%.1f GB resident), %d particles, %d projections of %d^2" % (nt, size, nt * size * size * 4 / 1e9, len(p2), m, n),
            "collective": "none (one rank)" if world == 1 else "one all-reduce of the accumulators after the last series",
            "median_deg_after": round(float(np.median(perr(results[0], parts_truth))), 3), "map_cc_vs_truth": round(float(np.corrcoef(fl.ravel(), vol.ravel())[0, 1]), 4)}


def csp_bench(ctx):
    """configs[3] at one tilt series per rank: P particles x 41 tilts of 128^2 boxes (resident), constrained refinement of the
    particle units (csp modes 2 / 5: three rotations + 3-D shift per particle, scored over its 41 projections).  Units shard
    over ranks with no collective (pyp_amd.dist.shard_units); every rank refines its own series here.  A step = one call of
    ppm_csp_refine over the series."""
    import torch
    from pyp_amd import host, synth
    from pyp_amd.abi import CSP_PARTICLES, CspCfg, RefineCfg
    a, rank, world, local, dev = (ctx[k] for k in ("a", "rank", "world", "local", "dev"))
    n, px, npart = 128, 2.0, a.csp_particles
    tl = np.linspace(-60, 60, 41)
    vol, stack, rows, parts, tilts = synth.make_tilt_series(n, npart, tl, pixel=px, snr=0.1, device=dev, seed=20240601 + rank)
    torch.cuda.synchronize()
    rng = np.random.default_rng(3 + rank)
    p2 = parts.copy()
    for i in range(len(p2)):
        Nm = synth.euler_matrix(-p2[i, 4], -p2[i, 5], -p2[i, 6])
        for k in range(3):
            Nm = Nm @ synth.rot_xyz(k, rng.normal(0, 2.0))
        p2[i, 4:7] = -synth.angles_from_matrix(Nm)
        p2[i, 1:4] += rng.normal(0, 1.0, 3)
    rows2 = synth.csp_rows_from_params(rows, parts, tilts, p2, tilts)
    cfg = RefineCfg.make(box=n, pixel_size=px, mask_radius=0.32 * n * px, res_high=px * n / (0.25 * n), res_signed_cc=30.0, global_search=0)
    host.lib.init(local)
    ref = host.Reference(vol, n / 2, device=local)
    cc = CspCfg.make(CSP_PARTICLES, tol_angle=(8, 8, 8), tol_shift=4.0)
    barrier = make_barrier(world)
    out = None
    for _ in range(a.warmup if a.no_side else max(a.warmup, 1)):
        out = ref.csp_refine(cfg, cc, stack, rows2, p2, tilts)
    host.profile(True, True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = ref.csp_refine(cfg, cc, stack, rows2, p2, tilts)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    prof = host.profile_report()
    host.profile(False, False)
    lc_csp = ref.last_counts()
    # config 4 puts four tilt series on every GPU: two of them in flight on two reference handles (calls on different handles may
    # run concurrently from different threads, include/ppm.h) fill the device while the other call's host side decides its next sweep
    two = None
    if rank == 0 and world == 1 and not a.no_side:
        import threading
        ref2 = host.Reference(vol, n / 2, device=local)
        ref2.csp_refine(cfg, cc, stack, rows2.copy(), p2.copy(), tilts.copy())
        outs = [None, None]

        errs = []

        def series(k, handle):
            try:
                for _ in range(a.steps):
                    outs[k] = handle.csp_refine(cfg, cc, stack, rows2.copy(), p2.copy(), tilts.copy())
            except Exception as e:          # noqa: BLE001 - reported in the line, the main figure stands
                errs.append(str(e)[:200])
        th = [threading.Thread(target=series, args=(k, h)) for k, h in enumerate((ref, ref2))]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt2 = time.perf_counter() - t1
        if errs or outs[0] is None or outs[1] is None:
            two = {"error": errs[0] if errs else "no result"}
        else:
            same = bool(np.array_equal(outs[0][1], out[1]) and np.array_equal(outs[1][1], out[1]))
            two = {"value": round(2 * len(rows) * a.steps / dt2, 1), "unit": "projections/s", "ms_per_series": round(dt2 / a.steps / 2 * 1e3, 2),
                   "results_equal_single_series_run": same}
        ref2.close()
    ref.close()
    pipe = None
    if not a.no_side:
        try:                    # the whole of config 4 per series: extraction -> refinement -> insertion, four series in flight
            pipe = csp_pipeline(ctx, vol, stack, rows2, p2, parts, tilts, cfg, cc, n, px)
        except Exception as e:          # noqa: BLE001
            pipe = {"error": str(e)[:300]}
    if rank != 0:
        return None

    def perr(x, y):
        return np.array([np.degrees(np.arccos(np.clip((np.trace(synth.euler_matrix(-u[4], -u[5], -u[6]).T @ synth.euler_matrix(-v[4], -v[5], -v[6])) - 1) / 2, -1, 1)))
                         for u, v in zip(x, y)])
    box384 = None
    if world == 1 and not a.no_side:
        try:                    # the box of the reference's tomography tutorial (docs/tutorials/tomo_empiar_10164.rst:454): 100 particles x 41 tilts of 384^2
            n3, px3, np3 = 384, 1.35, 100
            v3, st3, r3, pa3, ti3 = synth.make_tilt_series(n3, np3, tl, pixel=px3, snr=0.1, device=dev, seed=20240701)
            q3 = pa3.copy()
            for i in range(len(q3)):
                Nm = synth.euler_matrix(-q3[i, 4], -q3[i, 5], -q3[i, 6])
                for k in range(3):
                    Nm = Nm @ synth.rot_xyz(k, rng.normal(0, 2.0))
                q3[i, 4:7] = -synth.angles_from_matrix(Nm)
                q3[i, 1:4] += rng.normal(0, 1.0, 3)
            rr3 = synth.csp_rows_from_params(r3, pa3, ti3, q3, ti3)
            c3 = RefineCfg.make(box=n3, pixel_size=px3, mask_radius=0.32 * n3 * px3, res_high=px3 * n3 / (0.25 * n3), res_signed_cc=30.0, global_search=0)
            ref3 = host.Reference(v3, n3 / 2, device=local)
            ref3.csp_refine(c3, cc, st3, rr3, q3, ti3)
            host.profile(True, True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            o3 = ref3.csp_refine(c3, cc, st3, rr3, q3, ti3)
            dt3 = time.perf_counter() - t1
            pr3 = host.profile_report()
            host.profile(False, False)
            ref3.close()
            box384 = {"value": round(len(r3) / dt3, 1), "unit": "projections/s", "ms_per_step": round(dt3 * 1e3, 2),
                      "config": "%d particles x 41 tilts of 384^2 (%.1f GB resident), band r = 96 px, particle units" % (np3, len(r3) * n3 * n3 * 4 / 1e9),
                      "device_ms_per_step": {k2: round(v["ms"], 2) for k2, v in pr3.items() if v["launches"]},
                      "accuracy_vs_truth": {"median_deg_before": round(float(np.median(perr(q3, pa3))), 3), "median_deg_after": round(float(np.median(perr(o3[1], pa3))), 3)}}
            del st3
            torch.cuda.empty_cache()
        except Exception as e:          # noqa: BLE001
            box384 = {"error": str(e)[:300]}
    nproj = len(rows)
    blk = {"metric": "projections/sec constrained tilt-series refinement, 128^2 box", "value": round(world * nproj * a.steps / dt, 1), "unit": "projections/s",
           "particles_per_s": round(world * npart * a.steps / dt, 1), "n_gpus": world, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 2),
           "higher_is_better": True, "scaling": "weak", "dtype": "f32", "data": "synthetic",
           "config": {"workload": "one tilt series per GPU: %d particles x 41 tilts (+-60 deg), 128^2 boxes resident, particle units refined "
                                  "(3 rotations + 3-D shift, +-8 deg / +-4 px), band r = 32 px" % npart,
                      "projections_per_gpu": nproj, "parallelism": "unit-sharded x%d, no collective" % world},
           "device_ms_per_step": {k2: round(v["ms"] / a.steps, 2) for k2, v in prof.items() if v["launches"]},
           "device_busy_frac": round(sum(v["ms"] for v in prof.values()) * 1e-3 / dt, 3),
           "roofline": gather_roofline("k_csp_eval", "csp", prof["local"]["ms"], prof["local"]["launches"], lc_csp["samples_local"] * nproj * a.steps,
                                       nproj * a.steps * lc_csp["n_local"], "projections",
                                       "device-resident compass: %d sweeps per call, every sweep one launch over the usable projections of the refined units, no host wait between them" % lc_csp["n_local"]),
           "note": "compass search with its state on the device (k_csp_step_*): the device scores <= 13 candidate poses per projection and sweep (k_csp_eval, the sweep of "
                   "k_local: bound by the vector memory path like it, DESIGN.md 4b); the rest of a step is the host's row / unit tables before and the write-back after",
           "accuracy_vs_truth": {"median_deg_before": round(float(np.median(perr(p2, parts))), 3), "median_deg_after": round(float(np.median(perr(out[1], parts))), 3),
                                 "median_shift_px_after": round(float(np.median(np.linalg.norm(out[1][:, 1:4] - parts[:, 1:4], axis=1))), 3)}}
    if box384:
        blk["box384"] = box384
    if two:
        blk["two_series_in_flight"] = two
    if pipe:
        blk["pipeline"] = pipe
    if world == 1 and not a.no_cpu and a.cpu_seconds > 0:
        from oracle import oracle
        cores = host_cores()
        _omp_threads(cores)
        k = max(1, min(npart, cores))                           # a bounded sample of particle units with all their projections
        from pyp_amd.formats import cistem
        sel = np.where(np.isin(rows2[:, cistem.COL["PIND"]], p2[:k, 0]))[0]
        t0 = time.time()
        oref = oracle.Reference(vol, n / 2)
        cc1 = CspCfg.make(CSP_PARTICLES, tol_angle=(8, 8, 8), tol_shift=4.0, first=int(p2[0, 0]), last=int(p2[k - 1, 0]))
        t1 = time.time()
        orow, opart, _, _ = oracle.csp_refine(oref, cfg, cc1, stack.cpu().numpy(), rows2, p2, tilts)
        tc = time.time() - t1
        oref.close()
        blk["cpu_baseline"] = {"value": round(len(sel) / tc, 2), "unit": "projections/s", "cores": cores, "kind": "port",
                               "sample": "%d particle units (%d projections), %.1f s wall, OpenMP; reference preparation %.1f s excluded" % (k, len(sel), tc, t1 - t0)}
        # the same units in the timed GPU run: particle parameters (rotation, 3-D shift) and the scores of their projections
        dang = perr(out[1][:k], opart[:k])
        dsh = np.linalg.norm(out[1][:k, 1:4] - opart[:k, 1:4], axis=1)
        dsc = np.abs(out[0][sel, 14] - orow[sel, 14])
        blk["parity_vs_oracle"] = {"n_units": int(k), "n_projections": int(len(sel)), "max_deg": round(float(dang.max()), 4), "median_deg": round(float(np.median(dang)), 5),
                                   "max_shift_px": round(float(dsh.max()), 4), "median_shift_px": round(float(np.median(dsh)), 5),
                                   "max_abs_dSCORE": round(float(dsc.max()), 4), "tolerance": "0.1 deg / 0.5 px",
                                   "sample": "the first %d particle units of the timed series: GPU against oracle.csp_refine" % k}
    return blk


def sva_bench(ctx):
    """configs[4] at a resident batch per rank: V sub-volumes of 192^3 aligned to the reference (3DAVG refine mode: +-10 deg, +-10
    px, 7 compass iterations, missing wedge +-60 deg).  Sub-volumes shard over ranks by table rows with no collective.  A step =
    one call of ppm_sva_align over the batch (3-D FFT of every sub-volume + search)."""
    import torch
    from pyp_amd import host, synth
    from pyp_amd.abi import SvaCfg
    a, rank, world, local, dev = (ctx[k] for k in ("a", "rank", "world", "local", "dev"))
    n, nv = 192, a.sva_volumes
    vol, vols, poses, wedges = synth.make_subtomograms(n, nv, snr=0.1, device=dev, seed=20240701 + rank)
    torch.cuda.synchronize()
    cfg = SvaCfg.make(n, window=(0.33 * n, 0.33 * n, 0.33 * n), window_sigma=4.0, highpass=(0.05, 0.01), lowpass=(0.125, 0.05), tol_angle=10.0, tol_shift=10.0)
    start = synth.perturb_poses(poses, 3.0, 2.0)
    host.lib.init(local)
    ref = host.Reference(vol, n / 2, device=local)
    barrier = make_barrier(world)
    if not a.no_side:
        ref.sva_align(cfg, vols[:4], wedges[:4], start[:4])
    host.profile(True, True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out, sc = ref.sva_align(cfg, vols, wedges, start)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    prof = host.profile_report()
    host.profile(False, False)
    lc_sva = ref.last_counts()          # of the last timed alignment call
    # the protocol's global search (alignment_mode 0) from rotations anywhere on SO(3), on a bounded sample (outside the timed region above)
    glob = None
    if rank == 0 and not a.no_side:
        ng = min(nv, 64)
        rng = np.random.default_rng(12)
        gstart = poses[:ng].copy()
        for v in range(ng):
            R = synth.euler_matrix(rng.uniform(0, 360), np.degrees(np.arccos(rng.uniform(-1, 1))), rng.uniform(0, 360))
            gstart[v, :9] = (poses[v, :9].reshape(3, 3) @ R).ravel()
            gstart[v, 9:] += rng.normal(0, 2.0, 3)
        gcfg = SvaCfg.make(n, window=(0.33 * n, 0.33 * n, 0.33 * n), window_sigma=4.0, highpass=(0.05, 0.01), lowpass=(0.125, 0.05), tol_angle=10.0, tol_shift=10.0,
                           search_mode=1, global_step=15.0, n_candidates=25)
        torch.cuda.synchronize()
        try:                    # a side figure: its failure must not cost the line
            tg = time.perf_counter()
            gout, gsc = ref.sva_align(gcfg, vols[:ng], wedges[:ng], gstart)
            tg = time.perf_counter() - tg
            gerr = synth.pose_angle_error(gout, poses[:ng])
            glob = {"value": round(ng / tg, 1), "unit": "sub-volumes/s", "sample": "%d sub-volumes, random start rotations, shifts +-2 px; 15 deg grid (4 416 rotations), 25 candidates" % ng,
                    "median_deg_before": round(float(np.median(synth.pose_angle_error(gstart, poses[:ng]))), 1), "median_deg_after": round(float(np.median(gerr)), 3),
                    "frac_within_1deg": round(float((gerr < 1.0).mean()), 3)}
        except Exception as e:          # noqa: BLE001
            glob = {"error": str(e)[:200]}
    # ---- config 5's real regime: the sub-volumes live on the HOST (10 k x 192^3 = 283 GB) and stream through PCIe once per iteration -
    # alignment and averaging in one pass over page-locked host memory (ppm_sva_align_average: what sva.align_table / bin/sva_align call)
    streamed = None
    if rank == 0 and not a.no_side:
        pb = None
        try:
            pb = host.PinnedBuffer(vols.numel(), local)
            hv = pb.array.reshape(tuple(vols.shape))
            torch.from_numpy(hv).copy_(vols)
            torch.cuda.synchronize()
            acc_s = host.Accumulator(n, 1.0, "C1", device=local)
            ts = time.perf_counter()
            out_s, sc_s = ref.sva_align(cfg, hv, wedges, start, accumulator=acc_s)
            ts = time.perf_counter() - ts
            cnt_s = acc_s.counts()
            acc_s.close()
            streamed = {"value": round(nv / ts, 1), "unit": "sub-volumes/s", "host_to_device_GBps": round(nv * float(n) ** 3 * 4 / ts / 1e9, 1),
                        "poses_equal_resident_run": bool(np.array_equal(out_s, out)), "averaged": cnt_s,
                        "note": "%d sub-volumes in page-locked host memory: aligned AND averaged in one pass, every chunk uploaded once while the previous one is searched "
                                "(PCIe Gen5 x16 bounds this regime at ~1.9 k sub-volumes/s)" % nv}
        except Exception as e:          # noqa: BLE001
            streamed = {"error": str(e)[:300]}
        finally:
            if pb is not None:
                pb.close()
    ref.close()
    # ---- the averaging step of the iteration (ppm_sva_insert): the aligned sub-volumes into the half-map accumulators, then finalise
    avg_blk = None
    try:
        if a.no_side:
            raise RuntimeError("left out (--no-side)")
        from pyp_amd.abi import FinalCfg
        warm = host.Accumulator(n, 1.0, "C1", device=local)
        warm.sva_insert(cfg, vols[:8], wedges[:8], out[:8])
        warm.close()
        acc = host.Accumulator(n, 1.0, "C1", device=local)
        host.profile(True, True)
        barrier()
        t0 = time.perf_counter()
        acc.sva_insert(cfg, vols, wedges, out, index=np.arange(nv) + rank * nv)
        barrier()
        dta = max_over_ranks(time.perf_counter() - t0, world, dev)
        profa = host.profile_report()
        host.profile(False, False)
        if rank == 0:
            h1, h2, avg, st = acc.finalize(FinalCfg(molecular_mass_kda=0.0, inner_radius=0.0, outer_radius=0.0, mask_falloff=0.0))
            kk = np.arange(n) - n // 2
            zz, yy, xx = np.meshgrid(kk, kk, kk, indexing="ij")
            msk = (xx * xx + yy * yy + zz * zz) < (0.33 * n) ** 2
            cca = float(np.corrcoef(avg[msk], vol[msk])[0, 1])
            below = np.where(st[:, 3] < 0.5)[0]
            avg_blk = {"value": round(world * nv / dta, 1), "unit": "sub-volumes/s", "ms_per_sub_volume": round(dta / nv * 1e3, 4),
                       "device_ms_per_sub_volume": {"transforms": round(profa["prep"]["ms"] / nv, 4), "gather_insert": round(profa["insert"]["ms"] / nv, 4)},
                       "map_cc_vs_truth": round(cca, 4), "fsc_0.5_at_px": float(st[below[0], 1]) if len(below) else float(st[-1, 1]),
                       "align_plus_average_sub_volumes_per_s": round(world * nv / (dt / a.steps + dta), 1),
                       "note": "ppm_sva_insert on the %d resident sub-volumes at the poses the timed alignment returned (full 192^3 transforms, wedge-weighted gather "
                               "into the half-map accumulators); ppm_finalize outside the timing" % nv}
        acc_keep = acc
    except Exception as e:          # noqa: BLE001 - a side figure: reported, the alignment figure stands
        avg_blk, acc_keep = {"error": str(e)[:300]}, None
    if rank != 0:
        if acc_keep is not None:
            acc_keep.close()
        return None
    n3 = float(n) ** 3
    ms_prep = prof["prep"]["ms"] / (nv * a.steps)
    R_ = int(np.ceil(min(0.5, 0.125 + 3.7169 * 0.05) * n))          # band radius of the protocol's low-pass (weights >= 1e-3), Fourier pixels
    KX_, KY_ = min(n // 2 + 1, R_ + 1), min(n, 2 * R_ + 1)
    moved = 4.0 * n3 + 2 * 8.0 * n * n * KX_ + 4 * 8.0 * n * KX_ * KY_
    pre_traffic, pre_src = pmc_traffic_sum(pmc_latest("sva"), ("k_sva_x16", "k_sva_stats_sum", "k_sva_yz16", "k_sva_gather16"))
    blk = {"metric": "sub-volumes/sec sub-tomogram alignment, 192^3 box", "value": round(world * nv * a.steps / dt, 1), "unit": "sub-volumes/s",
           "n_gpus": world, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%d resident 192^3 sub-volumes per GPU (%.1f GB), +-10 deg / +-10 px, missing wedge +-60 deg, band 0.125 cycles/pixel" % (nv, nv * n3 * 4 / 1e9),
                      "sub_volumes_per_gpu": nv, "parallelism": "row-sharded x%d, no collective" % world},
           "device_ms_per_sub_volume": {"pre_processing": round(ms_prep, 3), "search": round(prof["local"]["ms"] / (nv * a.steps), 3)},
           "roofline": sva_eval_roofline(prof, lc_sva, nv, a.steps, wedges),
           "roofline_pre_processing": {"bound": "hbm", "kernel": "sub-volume pre-processing (k_sva_x16 + k_sva_stats_sum + two k_sva_yz16 passes; k_sva_gather16 only with PPM_SVA_FOLD=0)",
                        "achieved": round(4.0 * n3 / (ms_prep * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(4.0 * n3 / (ms_prep * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4), "traffic": pre_traffic, "traffic_source": pre_src,
                        "traffic_over_algorithmic": None if pre_traffic is None else round(pre_traffic / (4.0 * n3), 2),
                        "hbm_traffic_frac": None if pre_traffic is None else round(pre_traffic / (ms_prep * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                        "algorithmic_bytes": "4 n^3: the sub-volume read once",
                        "moved_bytes_model": {"bytes_per_sub_volume": round(moved), "GBps": round(moved / (ms_prep * 1e-3) / 1e9, 1),
                                              "note": "the volume is read once (the x pass gathers the statistics, the normalisation is applied at the samples); the pruned transform (kx <= R, |ky|, |kz| <= R) writes and "
                                                      "reads A[z][kx][y] once and B[kx][ky][z] twice; 32 sub-volumes per launch, so the work arrays (0.9 GB) "
                                                      "live in HBM; the passes are bound by their LDS transforms (DESIGN.md 9)"}},
           "pcie_bound_note": "config 5's 10 k sub-volumes (283 GB) stream from the host: 28 MB each, i.e. ~1.9 k sub-volumes/s at PCIe Gen5 rates",
           "accuracy_vs_truth": {"median_deg_before": round(float(np.median(synth.pose_angle_error(start, poses))), 3),
                                 "median_deg_after": round(float(np.median(synth.pose_angle_error(out, poses))), 3),
                                 "median_shift_px_after": round(float(np.median(np.linalg.norm(out[:, 9:] - poses[:, 9:], axis=1))), 3),
                                 "mean_score": round(float(sc.mean()), 4)},
           "global_search": glob, "average": avg_blk, "streamed": streamed}
    if world == 1 and not a.no_cpu and a.cpu_seconds > 0 and acc_keep is not None:
        try:                    # the average of the first sub-volumes by the oracle and by the HIP path, into fresh accumulators
            from oracle import oracle
            _omp_threads(host_cores())
            ka = min(nv, 4)
            ao, co = np.zeros(oracle.accum_floats(n), np.float32), np.zeros(2, np.int64)
            t0 = time.time()
            oracle.sva_insert(ao, co, cfg, vols[:ka].cpu().numpy(), wedges[:ka], out[:ka])
            tca = time.time() - t0
            ag = host.Accumulator(n, 1.0, "C1", device=local)
            ag.sva_insert(cfg, vols[:ka], wedges[:ka], out[:ka])
            gg, go = ag.download().reshape(-1, 3).astype(np.float64), ao.reshape(-1, 3).astype(np.float64)
            ag.close()
            same = gg[:, 2] == go[:, 2]
            avg_blk["parity_vs_oracle"] = {"n": int(ka), "rel_l2_values": float("%.3g" % (np.linalg.norm((gg - go)[same, :2]) / np.linalg.norm(go[:, :2]))),
                                           "voxels_with_other_weight_frac": float("%.3g" % (1.0 - same.mean())),
                                           "sample": "the first %d sub-volumes into fresh accumulators: ppm_sva_insert against oracle.sva_insert (%.1f s)" % (ka, tca)}
            avg_blk["cpu_baseline"] = {"value": round(ka / tca, 3), "unit": "sub-volumes/s", "cores": host_cores(), "kind": "port", "sample": "%d sub-volumes, %.1f s wall" % (ka, tca)}
        except Exception as e:          # noqa: BLE001
            avg_blk["parity_vs_oracle"] = {"error": str(e)[:300]}
    if acc_keep is not None:
        acc_keep.close()
    if world == 1 and not a.no_cpu and a.cpu_seconds > 0:
        from oracle import oracle
        cores = host_cores()
        _omp_threads(cores)
        k = min(nv, 16)
        t0 = time.time()
        oref = oracle.Reference(vol, n / 2)
        t1 = time.time()
        opose, osc, _ = oracle.sva_align(oref, cfg, vols[:k].cpu().numpy(), wedges[:k], start[:k])
        tc = time.time() - t1
        oref.close()
        dang = synth.pose_angle_error(out[:k], opose)
        dsh = np.linalg.norm(out[:k, 9:] - opose[:, 9:], axis=1)
        blk["parity_vs_oracle"] = {"n": int(k), "max_deg": round(float(dang.max()), 4), "median_deg": round(float(np.median(dang)), 5),
                                   "max_shift_px": round(float(dsh.max()), 4), "median_shift_px": round(float(np.median(dsh)), 5),
                                   "max_abs_dscore": round(float(np.abs(sc[:k] - osc).max()), 5), "tolerance": "0.1 deg / 0.5 px",
                                   "sample": "the first %d sub-volumes of the timed batch: GPU against oracle.sva_align" % k}
        blk["cpu_baseline"] = {"value": round(k / tc, 3), "unit": "sub-volumes/s", "cores": cores, "kind": "port",
                               "sample": "%d sub-volumes, %.1f s wall, OpenMP; reference preparation %.1f s excluded" % (k, tc, t1 - t0)}
    return blk


# --------------------------------------------------------------------------------------------- CPU legs
def host_cores():
    """CPU cores this process may really use: the affinity mask, capped by a cgroup CPU quota when one is set
    (a GPU box hands a share of the host's cores to each GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(round(quota))))
    if os.environ.get("PPM_CPU_CORES"):
        n = max(1, int(os.environ["PPM_CPU_CORES"]))
    return n


def pose_parity(want, got, px, what):
    """GPU rows against the oracle's rows of the same particles: the north-star tolerance is 0.1 deg / 0.5 px."""
    from pyp_amd import synth
    ang, shf = synth.angular_error_deg(want, got), synth.shift_error_px(want, got, px)
    dsc = np.abs(want[:, 14] - got[:, 14])
    return {"n": int(len(want)), "max_deg": round(float(ang.max()), 4), "median_deg": round(float(np.median(ang)), 5),
            "max_shift_px": round(float(shf.max()), 4), "median_shift_px": round(float(np.median(shf)), 5),
            "max_abs_dSCORE": round(float(dsc.max()), 4), "median_abs_dSCORE": round(float(np.median(dsc)), 5),
            "frac_within_0.1deg_0.5px": round(float(((ang < 0.1) & (shf < 0.5)).mean()), 4), "tolerance": "0.1 deg / 0.5 px (BASELINE.json north_star)",
            "sample": what}


def cpu_baseline(vol, stack, start_rows, cfg, N, seconds, gpu_rows=None, px=1.0):
    """The CPU oracle (kind "port": the reference binaries are absent) on bounded samples of the SAME stack,
    three ways: all host cores through OpenMP (the headline `value`), one thread, and P independent
    single-thread processes that each prepare the reference themselves - the reference's process model
    (one refine3d per particle range with OMP_NUM_THREADS=1, src/pyp/refine/frealign/frealign.py:3183).
    Returns (cpu_baseline block, parity block): the rows the OpenMP leg computes are compared with the GPU rows of the same
    particles of the timed run (`gpu_rows`) - the at-size parity figure of the line."""
    import ctypes
    import tempfile
    from oracle import oracle
    cores = host_cores()
    gomp = ctypes.CDLL("libgomp.so.1")
    imgs = stack[:min(len(stack), max(64 * cores, 256))].cpu().numpy()      # pool the timed sample is drawn from: the --cpu-seconds budget decides how many are used
    t0 = time.time()
    oref = oracle.Reference(vol, N / 2)
    t_prep = time.time() - t0
    legs = {}
    # ---- one thread, eight particles (+ its own reference preparation)
    mode = "oracle ccf_mode=1 (pruned separable transform of the shift window)"
    gomp.omp_set_num_threads(1)
    n1 = min(8, len(imgs))
    t0 = time.time()
    oracle.refine_batch(oref, cfg, imgs[:n1], start_rows[:n1], ccf_mode=1)
    t1 = (time.time() - t0) / n1                         # seconds per particle on one thread
    legs["single_thread"] = {"value": round(1.0 / t1, 4), "unit": "particles/s", "cores": 1,
                             "value_incl_reference_prep": round(n1 / (t1 * n1 + t_prep), 4),
                             "sample": "%d particles, %.1f s; reference preparation %.1f s; %s" % (n1, t1 * n1, t_prep, mode)}
    # ---- all cores, OpenMP over particles
    gomp.omp_set_num_threads(cores)
    n = cores
    t0 = time.time()
    orows, _ = oracle.refine_batch(oref, cfg, imgs[:n], start_rows[:n], ccf_mode=1)
    t2 = time.time() - t0
    if t2 < 0.5 * seconds and len(imgs) > n:
        n2 = max(cores, (int(min(len(imgs), max(256, n / t2 * seconds))) // cores) * cores)       # at least 256 particles (SURVEY 8d asks for a stable sample)
        t0 = time.time()
        orows, _ = oracle.refine_batch(oref, cfg, imgs[:n2], start_rows[:n2], ccf_mode=1)
        t2, n = time.time() - t0, n2
    legs["openmp_all_cores"] = {"value": round(n / t2, 3), "unit": "particles/s", "cores": cores,
                                "speedup_over_one_thread": round(n / t2 * t1, 1),
                                "sample": "%d particles, %.1f s wall, OpenMP over particles; reference preparation %.1f s excluded; %s" % (n, t2, t_prep, mode)}
    parity = None
    if gpu_rows is not None:
        parity = pose_parity(orows, gpu_rows[:n], px, "the first %d particles of the timed stack: rows of the timed GPU run against the oracle's OpenMP leg" % n)
    oref.close()
    # ---- P single-thread processes, one particle range each, reference prepared per process
    P = max(1, min(cores, 64))
    per = 1
    tmp = tempfile.mkdtemp(prefix="ppm_cpu_leg_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        np.save(os.path.join(tmp, "vol.npy"), vol)
        np.save(os.path.join(tmp, "imgs.npy"), imgs[:P * per])
        np.save(os.path.join(tmp, "rows.npy"), start_rows[:P * per])
        with open(os.path.join(tmp, "cfg.bin"), "wb") as f:
            f.write(bytes(cfg))
        env = dict(os.environ, OMP_NUM_THREADS="1", NCPUS="1")
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_leg.py"), tmp, str(i * per), str((i + 1) * per)], env=env)
                 for i in range(P)]
        rcs = [p.wait() for p in procs]
        t3 = time.time() - t0
        if any(rcs):
            print("WARNING: %d of %d CPU-leg processes failed" % (sum(1 for r in rcs if r), P), file=sys.stderr)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    if all(r == 0 for r in rcs):
        legs["processes_single_thread"] = {"value": round(P * per / t3, 3), "unit": "particles/s", "cores": P,
                                           "sample": "%d processes x %d particle(s), OMP_NUM_THREADS=1, each prepares the reference itself "
                                                     "(frealign.py:3183 process model), %.1f s wall" % (P, per, t3)}
    best = legs["openmp_all_cores"]
    return {"value": best["value"], "unit": "particles/s", "cores": cores, "kind": "port", "sample": best["sample"], "legs": legs}, parity


if __name__ == "__main__":
    sys.exit(main())
