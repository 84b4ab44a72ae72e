"""First-contact diagnostics on the GPU box: compares each stage against the CPU oracle and prints numbers."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyp_amd import synth, host
from pyp_amd.abi import RefineCfg, ReconCfg, FinalCfg
from oracle import oracle

def main():
    N = int(os.environ.get("N", 64)); M = int(os.environ.get("M", 16))
    px = 2.0
    vol, stack, rows = synth.make_dataset(N, M, pixel=px, snr=0.1)
    imgs = stack.numpy()
    oref = oracle.Reference(vol, N / 2)
    t = time.time(); gref = host.Reference(vol, N / 2); print("gpu reference", time.time() - t)
    base = dict(box=N, pixel_size=px, mask_radius=0.4 * N * px, res_high=px * N / 24.0, res_search=px * N / 10.0,
                search_range_x=12.0, search_range_y=12.0, res_signed_cc=30.0)
    # 1. score only
    cfg = RefineCfg.make(**base, global_search=0, local_refine=0)
    so = oracle.score_batch(oref, cfg, imgs, rows)
    out = gref.refine(cfg, imgs, rows)
    print("score-only: oracle", so[:4], "gpu", out[:4, 14] / 100, "maxdiff", np.abs(so - out[:, 14] / 100).max())
    # 2. local only
    pr = synth.perturb_rows(rows, 2.0, 1.0, px)
    cfg = RefineCfg.make(**base, global_search=0, local_refine=1)
    oo, _ = oracle.refine_batch(oref, cfg, imgs, pr)
    t = time.time(); og = gref.refine(cfg, imgs, pr); print("gpu local", time.time() - t)
    print("local: ang diff", synth.angular_error_deg(oo, og).round(4), "shift diff", synth.shift_error_px(oo, og, px).round(4))
    print("       score diff", np.abs(oo[:, 14] - og[:, 14]).max())
    # 3. global only
    cfg = RefineCfg.make(**base, global_search=1, local_refine=0)
    oo, _ = oracle.refine_batch(oref, cfg, imgs, rows)
    t = time.time(); og = gref.refine(cfg, imgs, rows); print("gpu global", time.time() - t)
    print("global: ang diff", synth.angular_error_deg(oo, og).round(4), "shift diff", synth.shift_error_px(oo, og, px).round(4))
    print("       score oracle", oo[:4, 14], "gpu", og[:4, 14])
    # 4. full
    cfg = RefineCfg.make(**base, global_search=1, local_refine=1)
    oo, cnt = oracle.refine_batch(oref, cfg, imgs, rows)
    t = time.time(); og = gref.refine(cfg, imgs, rows); print("gpu full", time.time() - t, gref.last_counts(), cnt)
    print("full: ang diff", synth.angular_error_deg(oo, og).round(4), "shift diff", synth.shift_error_px(oo, og, px).round(4))
    print("      vs truth gpu", synth.angular_error_deg(og, rows).round(2), "oracle", synth.angular_error_deg(oo, rows).round(2))
    # 5. insertion
    rc = ReconCfg(box=N, pixel_size=px, res_limit=2 * px, score_weight_bfactor=2.0, score_average=20.0, score_threshold=0, normalize=1, invert=0,
                  split_by_pind=0, mask_radius=0.4 * N * px)
    rr = rows.copy(); rr[:, 14] = np.linspace(10, 30, M)
    acc = np.zeros(oracle.accum_floats(N), dtype=np.float32); counts = np.zeros(2, dtype=np.int64)
    oracle.insert_batch(acc, counts, rc, "C2", imgs, rr)
    ga = host.Accumulator(N, px, "C2")
    ga.insert(rc, imgs, rr)
    gacc = ga.download()
    print("insert counts", counts, ga.counts(), "rel L2 diff", np.linalg.norm(gacc - acc) / np.linalg.norm(acc), "max abs", np.abs(gacc - acc).max(), np.abs(acc).max())
    fc = FinalCfg(molecular_mass_kda=300.0, inner_radius=0, outer_radius=0.45 * N * px, mask_falloff=0)
    h1, h2, fl, st = oracle.finalize(acc, N, px, fc)
    g1, g2, gf, gs = ga.finalize(fc)
    for nm, a, b in (("half1", h1, g1), ("half2", h2, g2), ("filt", fl, gf)):
        print("finalize", nm, "rel L2", np.linalg.norm(a - b) / np.linalg.norm(a))
    print("stats max diff", np.abs(st - gs).max(axis=0))

if __name__ == "__main__":
    main()
