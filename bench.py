#!/usr/bin/env python3
"""bench.py — particles/s of per-particle projection matching (BASELINE.json metric) on N MI355X.

A "step" is one pass of the hot path (pre-processing FFT + CTF tables, global grid search,
top-hit + final local refinement: one ppm_refine_batch call) over the rank's synthetic
particle stack, which is resident in HBM before the timed region starts.

Workload at N=1 = BASELINE.json configs[1]: "SPA global search: 100k 256^2 particles,
15 deg angular step, 1 MI355X" (search band r = 64 Fourier px).  Particles
shard across ranks with no data-path collective (weak scaling: --particles is per GPU).

Contract: python bench.py --gpus N --steps K --warmup W   (torchrun launches N ranks for N > 1)
prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--particles", type=int, default=100000, help="particles per GPU per step (resident stack)")
    ap.add_argument("--box", type=int, default=256)
    ap.add_argument("--band", type=float, default=64.0, help="search / refinement band limit, Fourier pixels")
    ap.add_argument("--angular-step", type=float, default=15.0)
    ap.add_argument("--unique", type=int, default=512, help="distinct clean projections (each particle gets fresh noise)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target wall time of the CPU oracle sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", choices=["refine", "reconstruct"], default="refine",
                    help="refine = BASELINE.json configs[1] (the headline metric); reconstruct = configs[2]: Fourier insertion into "
                         "half-map accumulators + one all-reduce over the ranks")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "PPM_FORCE_DEVICE" in os.environ:          # rehearsal of the N > 1 path on a one-GPU box (with PPM_DIST_BACKEND=gloo)
        local = int(os.environ["PPM_FORCE_DEVICE"])
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        print("ERROR: bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1:
        backend = os.environ.get("PPM_DIST_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from pyp_amd import host, synth
    from pyp_amd.abi import RefineCfg

    N, M, px = a.box, a.particles, 1.0
    dev = torch.device("cuda", local)
    # ---- synthetic inputs: phantom, poses, CTF, SNR 0.05; rank r gets its own poses/noise
    vol = synth.phantom(N)
    _, stack, rows = synth.make_dataset(N, M, pixel=px, snr=0.05, vol=vol, device=dev, unique=min(a.unique, M),
                                        seed_poses=synth.SEED_POSES + rank, seed_noise=synth.SEED_NOISE + rank, batch=32)
    torch.cuda.synchronize()
    res = px * N / a.band
    cfg = RefineCfg.make(box=N, pixel_size=px, mask_radius=0.32 * N * px, res_high=res, res_search=res, res_low=0.0,
                         angular_step=a.angular_step, top_hits=20, search_range_x=6.0 * px, search_range_y=6.0 * px,
                         res_signed_cc=30.0, molecular_mass_kda=500.0)
    if a.workload == "reconstruct":
        return reconstruct_bench(a, rank, world, local, dev, vol, stack, rows, N, M, px)
    t0 = time.time()
    ref = host.Reference(vol, N / 2, device=local)
    t_refprep = time.time() - t0
    start_rows = synth.cistem.default_rows(M, px, 300.0, 2.7, 0.07)      # from-scratch rows: the search ignores the poses
    for c in ("DEFOCUS_1", "DEFOCUS_2", "DEFOCUS_ANGLE"):
        start_rows[:, synth.cistem.COL[c]] = rows[:, synth.cistem.COL[c]]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        host.lib.load().ppm_device_sync()

    out = None
    for _ in range(a.warmup):
        out = ref.refine(cfg, stack, start_rows)
    host.profile(True, True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = ref.refine(cfg, stack, start_rows)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = host.profile_report()
    host.profile(False, False)
    counts = ref.last_counts()

    if rank == 0:
        total = world * M * a.steps
        value = total / dt
        # ---- roofline of the dominant kernel: algorithmic bytes per launch / measured launch time (HIP events
        # recorded by the library on its own stream around every launch).  SURVEY.md §8(d) streaming model:
        # 8 S(r) bytes per orientation evaluated (+ 4 N^2 + 128 per particle for the whole path).
        S_g = counts["samples_global"]
        launches_g = max(prof["global"]["launches"], 1)
        per_launch_particles = M * a.steps / launches_g
        bytes_g = per_launch_particles * counts["n_global"] * 8.0 * S_g
        ms_g = prof["global"]["ms"] / launches_g
        bytes_l_total = M * a.steps * 8.0 * counts["samples_local"]          # samples_local = sum over local evaluations
        ms_l_total = max(prof["local"]["ms"], 1e-9)
        dom = "global" if prof["global"]["ms"] >= prof["local"]["ms"] else "local"
        if dom == "global":
            achieved = bytes_g / (ms_g * 1e-3) / 1e9
            kname, kms, kbytes = "k_global", ms_g, bytes_g
        else:
            nl = max(prof["local"]["launches"], 1)
            achieved = bytes_l_total / (ms_l_total * 1e-3) / 1e9
            kname, kms, kbytes = "k_local", ms_l_total / nl, bytes_l_total / nl
        b_pm = 4.0 * N * N + counts["n_global"] * 8.0 * S_g + 8.0 * counts["samples_local"] + 128
        traffic, traffic_src = pmc_traffic(kname, per_launch_particles if dom == "global" else M * a.steps / max(prof["local"]["launches"], 1))
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(kms, 3),
                "algorithmic_bytes_per_launch": kbytes,
                "note": "streaming-model bytes (8 S(r) per orientation); the slice bank is served from L2 / Infinity Cache and each "
                        "stored slice serves psi and psi+180, so frac can exceed the HBM-only ceiling; the kernel is fp32-VALU bound",
                "path_bytes_per_particle": b_pm, "path_achieved_GBps": round(b_pm * M * a.steps / dt / 1e9, 1),
                "path_frac": round(b_pm * M * a.steps / dt / 8e12, 4)}
        if dom == "global":
            # secondary: fp32 vector rate of k_global.  Per slice row pair and lane: 24 FMA (shift rows), 2 x 3 for the norm,
            # 8 mul + 12 add for the (A, Bq) even/odd parts; plus ~800 ops per lane and slice for the window reduction.
            R = 3
            flops_slice = 64 * 64 * (2 * 8 * R + 12 + 20) + 64 * 800
            flops = per_launch_particles * (counts["n_global"] / 2) * flops_slice
            roof["k_global_fp32_TFLOPs"] = round(flops / (ms_g * 1e-3) / 1e12, 1)
            roof["k_global_fp32_frac_of_157"] = round(flops / (ms_g * 1e-3) / 157.3e12, 3)
        # ---- accuracy of what was timed (vs the synthetic ground truth), first 2000 particles
        k = min(M, 2000)
        ang = synth.angular_error_deg(out[:k], rows[:k])
        shf = synth.shift_error_px(out[:k], rows[:k], px)
        line = {
            "metric": "particles/sec projection-matching, 256^2 box", "value": round(value, 1), "unit": "particles/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SPA global search: %dk %d^2 particles/GPU, %g deg angular step, band r=%g px, top-20 hits refined"
                       % (M // 1000, N, a.angular_step, a.band), "particles_per_gpu": M, "box": N, "orientations": counts["n_global"],
                       "local_evaluations": counts["n_local"], "parallelism": "particle-sharded x%d" % world},
            "roofline": roof,
            "kernels_ms": {k2: round(v["ms"], 2) for k2, v in prof.items() if v["launches"]},
            "reference_prep_s": round(t_refprep, 3),
            "accuracy_vs_truth": {"median_deg": round(float(np.median(ang)), 3), "frac_within_2deg": round(float((ang < 2).mean()), 3),
                                  "median_shift_px": round(float(np.median(shf)), 3)},
        }
        if not a.no_cpu and a.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(vol, stack, start_rows, cfg, N, a.cpu_seconds)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel, particles_per_launch, summary="r01_final5_pmc_traffic_refine_8k.json", profiled=8000):
    """HBM-side bytes per launch of `kernel` from a committed rocprofv3 --pmc summary (scripts/pmc_traffic.sh: FETCH_SIZE and
    WRITE_SIZE in separate passes; units of 1 KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), scaled
    from the profiled particle count to this launch.  None when no summary is committed."""
    path = os.path.join(ROOT, "profiles", summary)
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    key = [k for k in d if kernel in k]
    if not key or "FETCH_SIZE" not in d[key[0]] or "WRITE_SIZE" not in d[key[0]]:
        return None, None
    e = d[key[0]]
    per_particle = (2.0 * e["FETCH_SIZE"]["sum"] + e["WRITE_SIZE"]["sum"]) * 1024.0 / profiled
    return per_particle * particles_per_launch, "profiles/%s (%d particles, FETCH_SIZE x2 + WRITE_SIZE, KB)" % (summary, profiled)


def reconstruct_bench(a, rank, world, local, dev, vol, stack, rows, N, M, px):
    """configs[2]: every rank inserts its particles into private accumulators (a torch tensor handed to the
    library), then ONE all-reduce (sum, f32) over RCCL; finalisation on rank 0 is outside the timed region."""
    import torch
    import torch.distributed as dist
    from pyp_amd import dist as pdist
    from pyp_amd import host
    from pyp_amd.abi import FinalCfg, ReconCfg
    nfl = int(host.lib.load().ppm_accum_floats(N))
    host.lib.init(local)
    acc_t = torch.zeros(nfl, dtype=torch.float32, device=dev)
    acc = host.Accumulator(N, px, "C1", device=local, ext_tensor=acc_t)
    rows = rows.copy()
    rows[:, 0] += rank * M                 # global positions: half assignment must not depend on the rank count
    rc = ReconCfg(box=N, pixel_size=px, res_limit=2 * px, score_weight_bfactor=0.0, score_average=0.0, score_threshold=0.0,
                  normalize=1, invert=0, split_by_pind=0, mask_radius=0.32 * N * px)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        host.lib.load().ppm_device_sync()

    def step():
        acc_t.zero_()
        torch.cuda.synchronize()
        acc.set_counts(0, 0)
        acc.insert(rc, stack, rows)
        return pdist.reduce_accumulators(acc_t, acc.counts())[1]

    counts = None
    for _ in range(a.warmup):
        counts = step()
    host.profile(True, True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        counts = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = host.profile_report()
    host.profile(False, False)
    if rank == 0:
        S = int(np.floor(np.pi * (N / 2) ** 2 / 2))
        b_ins = 4.0 * N * N + S * 192.0
        nl = max(prof["insert"]["launches"], 1)
        ms = prof["insert"]["ms"] / nl
        achieved = (M * a.steps / nl) * (S * 192.0) / (ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic("k_insert_bricks", M * a.steps / nl, "r01_final5_pmc_traffic_reconstruct_16k.json", 16000)
        acc.set_counts(counts[0], counts[1])
        h1, h2, fl, stats = acc.finalize(FinalCfg(molecular_mass_kda=500.0, inner_radius=0.0, outer_radius=0.45 * N * px, mask_falloff=0.0))
        cc = float(np.corrcoef(fl.ravel(), vol.ravel())[0, 1])
        line = {"metric": "particles/sec Fourier insertion, 256^2 box", "value": round(world * M * a.steps / dt, 1), "unit": "particles/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "3D reconstruction: Fourier-insert %dk %d^2 particles/GPU -> %d^3 half-maps, C1, one all-reduce"
                           % (M // 1000, N, N), "particles_per_gpu": M, "box": N, "parallelism": "particle-sharded x%d" % world},
                "roofline": {"bound": "hbm", "kernel": "k_insert_bricks", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                             "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(ms, 3),
                             "note": "algorithmic bytes (SURVEY 8d) = S(N/2) x 8 taps x 12 B x 2 (read-modify-write) per particle, i.e. what a "
                                     "scatter into HBM would move; k_insert_bricks keeps 16^3-voxel bricks of the accumulator in LDS "
                                     "(64-bit fixed point, ds_add_u64) and touches HBM once per brick and launch, so its real HBM traffic "
                                     "is far below that figure (see traffic) and the kernel is VALU / LDS-atomic bound (profiles/r01_bricks_pmc_reconstruct_8k.json)",
                             "path_bytes_per_particle": b_ins},
                "kernels_ms": {k2: round(v["ms"], 2) for k2, v in prof.items() if v["launches"]},
                "map_cc_vs_truth": round(cc, 4), "fsc_at_half_nyquist": round(float(stats[N // 4 - 1, 3]), 4)}
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(vol, stack, start_rows, cfg, N, seconds):
    """The CPU oracle (kind "port": the reference binaries are absent) on a bounded sample of
    the SAME stack, all host cores via OpenMP, reference preparation excluded like on the GPU side."""
    from oracle import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    t0 = time.time()
    oref = oracle.Reference(vol, N / 2)
    t_prep = time.time() - t0
    n = cores
    imgs = stack[:max(4 * cores, 8)].cpu().numpy()
    t0 = time.time()
    oracle.refine_batch(oref, cfg, imgs[:n], start_rows[:n], ccf_mode=1)
    t1 = time.time() - t0
    rate = n / t1
    if t1 < 0.5 * seconds and len(imgs) > n:     # extend the sample towards the time budget
        n2 = int(min(len(imgs), max(n, rate * seconds)))
        n2 = max(cores, (n2 // cores) * cores)
        t0 = time.time()
        oracle.refine_batch(oref, cfg, imgs[:n2], start_rows[:n2], ccf_mode=1)
        t1 = time.time() - t0
        n = n2
        rate = n / t1
    return {"value": round(rate, 3), "unit": "particles/s", "cores": cores, "kind": "port",
            "sample": "%d particles of the same stack, %.1f s wall, OpenMP over particles; reference prep %.1f s excluded" % (n, t1, t_prep)}


if __name__ == "__main__":
    main()
