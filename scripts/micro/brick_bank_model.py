"""Bank-conflict model of k_insert_bricks' LDS atomics (ds_add_u64): groups of 64 queued hits of random slice planes through a 16^3
brick, 24 adds per hit; cost of one wave instruction = sum over the two half-waves of the largest number of lanes that fall on one
bank pair (64 banks x 4 B, an 8-byte cell takes two).  Compares cell layouts (strides in 8-byte cells).  Host-side study only."""
import numpy as np, sys
rng = np.random.default_rng(1)
BE = 16; BH = BE + 1

def rot():
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w)], [2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], [2*(x*z-y*w), 2*(y*z+x*w), 1-2*(x*x+y*y)]])

def hits_of_cut():
    """hits of one (particle, brick): base cells in queue order (rows of ky, consecutive kx)"""
    while True:
        M = rot(); c0, c1 = M[:, 0], M[:, 1]
        lo = rng.integers(-7, 7, 3) * BE; lo[0] = abs(lo[0])
        ctr = lo + BE / 2
        ka, kb = c0 @ ctr, c1 @ ctr
        ks = np.arange(int(ka) - 20, int(ka) + 21); kt = np.arange(int(kb) - 20, int(kb) + 21)
        KX, KY = np.meshgrid(ks, kt)          # rows = ky
        P = KX[..., None] * c0 + KY[..., None] * c1
        base = np.floor(P).astype(int) - lo
        ok = ((base >= 0) & (base < BE)).all(-1) & (KX >= 0) & (KX * KX + KY * KY < 127 * 127)
        if ok.sum() >= 32:
            return base[ok]                   # row-major order = queue order

def cost(cells, layout):
    SX, SY, SZ, comp = layout
    tot = 0; n = 0
    for g0 in range(0, len(cells) - 63, 64):
        g = cells[g0:g0 + 64]
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    for c in range(3):
                        a = (g[:, 2] + dz) * SZ + (g[:, 1] + dy) * SY + (g[:, 0] + dx) * SX + c * comp
                        slot = a % 32
                        for h in (slice(0, 32), slice(32, 64)):
                            # lanes on the same slot serialise (same address too: atomics)
                            tot += np.bincount(slot[h], minlength=32).max()
                        n += 1
    return tot, n

layouts = {"interleaved SY=52 SZ=887 (now)": (3, 52, 887, 1), "interleaved unpadded 51/867": (3, 51, 867, 1), "interleaved 53/903": (3, 53, 903, 1),
           "interleaved 51/869": (3, 51, 869, 1), "interleaved 55/937": (3, 55, 937, 1),
           "planes x1 y17 z289 (+4913)": (1, 17, 289, 4913), "planes y17 z291 comp 4947+1": (1, 17, 291, 4948), "planes y19 z325": (1, 19, 325, 5527),
           "planes y17 z293": (1, 17, 293, 4982), "planes y18 z307": (1, 18, 307, 5220), "planes y21 z359": (1, 21, 359, 6105)}
cuts = []
tot_hits = 0
while tot_hits < 40000:
    c = hits_of_cut(); cuts.append(c); tot_hits += len(c)
allc = np.concatenate(cuts)           # the kernel's queue crosses rows but not cuts; close enough for a bank model
for name, L in layouts.items():
    t = n = 0
    for c in cuts:
        a, b = cost(c, L); t += a; n += b
    print("%-36s passes per ds_add_u64: %.2f (ideal 2.00)" % (name, t / max(n, 1)))
