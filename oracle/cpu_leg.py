"""One single-thread `refine3d`-like process of bench.py's CPU baseline (the reference's process model: one process per
particle range, OMP_NUM_THREADS=1, each preparing the reference itself; src/pyp/refine/frealign/frealign.py:3183).

TEST / BENCH INFRASTRUCTURE ONLY (oracle/): usage  cpu_leg.py <dir> <first> <last>
<dir> holds vol.npy, imgs.npy, rows.npy and cfg.bin (the raw ppm_refine_cfg struct)."""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:] = [os.path.dirname(_HERE)] + [p for p in sys.path if os.path.abspath(p or ".") != _HERE]      # `oracle` must be the package


def main():
    d, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from oracle import oracle
    from pyp_amd.abi import RefineCfg
    cfg = RefineCfg.from_buffer_copy(open(os.path.join(d, "cfg.bin"), "rb").read())
    vol = np.load(os.path.join(d, "vol.npy"))
    imgs = np.load(os.path.join(d, "imgs.npy"), mmap_mode="r")[first:last]
    rows = np.load(os.path.join(d, "rows.npy"))[first:last]
    ref = oracle.Reference(vol, vol.shape[0] / 2)
    oracle.refine_batch(ref, cfg, np.ascontiguousarray(imgs), rows, ccf_mode=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
