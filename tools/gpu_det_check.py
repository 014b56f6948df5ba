"""Run-to-run determinism of the reduced solve (dataflow launches, LDS-DMA K loops, back substitution): the same system solved
repeatedly must give bit-identical x -- a stale or early read of another workgroup's tile would show up here.
  gpurun -- python tools/gpu_det_check.py"""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from camera_calibration_amd import engine as eng
from oracle import oracle as orc
def system(n_blocks, dd, seed):
    rng = np.random.default_rng(seed)
    s = orc.System(6, n_blocks, dd)
    A = rng.normal(size=(dd, 768))
    s.dense_H[:] = np.triu(A @ A.T + dd * np.eye(dd))
    s.off_diag_H[:] = rng.normal(size=(6 * n_blocks, dd)) * 0.1
    for b in range(n_blocks):
        M = rng.normal(size=(6, 6)); s.block_diag_H[b] = np.triu(M @ M.T + 6 * np.eye(6))
    s.block_diag_b[:] = rng.normal(size=6 * n_blocks); s.dense_b[:] = rng.normal(size=dd)
    return s
for dd, reps in ((3500, 30), (7000, 30), (12525, 40), (22617, 8)):
    s = system(12, dd, dd)
    xs = [eng.schur_solve(s.block_diag_H, s.off_diag_H, s.dense_H, s.block_diag_b, s.dense_b) for _ in range(reps)]
    dev = max(np.abs(x - xs[0]).max() for x in xs[1:]) / np.abs(xs[0]).max()
    print("D", dd, "reps", reps, "max run-to-run |dx|/|x|max", dev, flush=True)
