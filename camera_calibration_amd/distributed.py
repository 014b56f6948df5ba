"""Image sharding across GPUs (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Every residual touches exactly one imageset pose, so with the poses eliminated the pose blocks D_i,
their strips B_i and b_i are local to the rank that owns imageset i, and the reduced system is a plain
sum over ranks:

    S = H_dd + lambda I - sum_i B_i^T (D_i + lambda I)^-1 B_i ,   s = b_d - sum_i B_i^T (D_i + lambda I)^-1 b_i

Each rank builds its partial (H_dd,local - sum_{i local} ...) on the device, ONE all-reduce per
Gauss-Newton step sums the contiguous buffer (matrix + right-hand side), lambda is added once after the
reduction, the factorisation is replicated and the pose back-substitution stays local.  Scalars
(costs, CostIsSmallerThan sums, the diagonal sum for the initial lambda) use 8-double all-reduces.
The LM control flow lives in cba_step (C++); this module only supplies the all-reduce callback.
"""
from __future__ import annotations

import ctypes
from typing import List, Tuple

import numpy as np


def shard_images(obs_per_image: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous imageset ranges per rank, balanced by observation count (all cameras of an imageset
    stay together because they share the rig pose)."""
    n = len(obs_per_image)
    csum = np.concatenate([[0], np.cumsum(obs_per_image, dtype=np.int64)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(csum, target))
        k = min(max(k, bounds[-1]), n)
        bounds.append(k)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def make_allreduce(big_buffer, local_rank: int):
    """Returns callback(ptr, count) -> 0 that sums a DEVICE fp64 buffer over all ranks.

    `big_buffer` is the torch tensor handed to the engine as its reduce buffer (so the reduced system
    is all-reduced in place, no copy); small library-owned scalar buffers are staged through a tiny
    torch tensor with device-to-device copies."""
    import torch
    import torch.distributed as dist

    hip = ctypes.CDLL("libamdhip64.so.7")  # resolves to the runtime already loaded by torch
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemcpy.restype = ctypes.c_int
    HIP_MEMCPY_D2D = 3
    staging = torch.zeros(64, dtype=torch.float64, device=f"cuda:{local_rank}")
    big_ptr = big_buffer.data_ptr() if big_buffer is not None else 0

    def allreduce(ptr: int, count: int) -> int:
        if big_buffer is not None and ptr == big_ptr:
            dist.all_reduce(big_buffer[:count])
            torch.cuda.synchronize()
            return 0
        if count > staging.numel():
            return 1
        if hip.hipMemcpy(staging.data_ptr(), ptr, count * 8, HIP_MEMCPY_D2D) != 0:
            return 1
        dist.all_reduce(staging[:count])
        torch.cuda.synchronize()
        if hip.hipMemcpy(ptr, staging.data_ptr(), count * 8, HIP_MEMCPY_D2D) != 0:
            return 1
        return 0

    return allreduce


# ---------------------------------------------------------------------------------------------------
# host-side statement of the sharded reduction (numpy) -- used by the CPU (gloo) tests to pin the
# algebra the device path implements: partial reduced systems add up, lambda enters once.
# ---------------------------------------------------------------------------------------------------
def local_reduced_system(block_diag_H: np.ndarray, off_diag_H: np.ndarray, dense_H: np.ndarray,
                         block_diag_b: np.ndarray, dense_b: np.ndarray, lam: float):
    """Partial reduced system of one rank, WITHOUT lambda on the dense diagonal.
    Inputs in the reference layout (upper triangles).  Returns (S_partial_upper, s_partial, Dinv_B, Dinv_b)."""
    nb, bs, _ = block_diag_H.shape
    dd = dense_H.shape[0]
    S = np.triu(dense_H).copy()
    s = dense_b.copy()
    Dinv_B = np.zeros_like(off_diag_H)
    Dinv_b = np.zeros_like(block_diag_b)
    for i in range(nb):
        D = np.triu(block_diag_H[i]) + np.triu(block_diag_H[i], 1).T + lam * np.eye(bs)
        Dinv = np.linalg.inv(D)
        B = off_diag_H[i * bs:(i + 1) * bs]
        W = Dinv @ B
        Dinv_B[i * bs:(i + 1) * bs] = W
        Dinv_b[i * bs:(i + 1) * bs] = Dinv @ block_diag_b[i * bs:(i + 1) * bs]
        S -= np.triu(B.T @ W)
        s -= B.T @ Dinv_b[i * bs:(i + 1) * bs]
    return S, s, Dinv_B, Dinv_b


def solve_reduced(S_upper_sum: np.ndarray, s_sum: np.ndarray, lam: float) -> np.ndarray:
    dd = S_upper_sum.shape[0]
    S = np.triu(S_upper_sum) + np.triu(S_upper_sum, 1).T + lam * np.eye(dd)
    return np.linalg.solve(S, s_sum)
