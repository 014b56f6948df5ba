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
        # the engine's stream is idle when this is called (include/cba.h); the collective is ordered on torch's current
        # stream, and only that stream is waited for -- no device-wide synchronisation
        if big_buffer is not None and ptr == big_ptr:
            dist.all_reduce(big_buffer[:count])
            torch.cuda.current_stream().synchronize()
            return 0
        if count > staging.numel():
            return 1
        if hip.hipMemcpy(staging.data_ptr(), ptr, count * 8, HIP_MEMCPY_D2D) != 0:
            return 1
        dist.all_reduce(staging[:count])
        torch.cuda.current_stream().synchronize()
        if hip.hipMemcpy(ptr, staging.data_ptr(), count * 8, HIP_MEMCPY_D2D) != 0:
            return 1
        return 0

    return allreduce


def make_allreduce_host_staged():
    """callback(ptr, count) that sums a DEVICE fp64 buffer over all ranks through HOST memory with whatever process group is
    initialised (gloo).  For tests that run several ranks on ONE GPU (RCCL refuses two ranks on one device) and for
    bring-up on machines without a GPU interconnect; the production path is make_allreduce (RCCL over xGMI) or the native
    callback of libcalib_ba_rccl.so."""
    import torch
    import torch.distributed as dist

    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemcpy.restype = ctypes.c_int
    D2H, H2D = 2, 1

    def allreduce(ptr: int, count: int) -> int:
        host = np.empty(count, dtype=np.float64)
        if hip.hipMemcpy(host.ctypes.data, ptr, count * 8, D2H) != 0:
            return 1
        t = torch.from_numpy(host)
        dist.all_reduce(t)
        if hip.hipMemcpy(ptr, host.ctypes.data, count * 8, H2D) != 0:
            return 1
        return 0

    return allreduce


class NativeRccl:
    """libcalib_ba_rccl.so (include/cba_rccl.h): the all-reduce callback a C++ host uses, bound for Python callers.
    `fn` / `user` go straight into cba_config.allreduce / allreduce_user -- no Python frame on the reduction path."""

    def __init__(self, rank: int, world: int, id_file: str, device: int):
        import os
        here = os.path.dirname(os.path.abspath(__file__))
        self.lib = ctypes.CDLL(os.path.join(here, "libcalib_ba_rccl.so"))
        self.lib.cba_rccl_create_via_file.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        self.lib.cba_rccl_destroy.argtypes = [ctypes.c_void_p]
        self.lib.cba_rccl_last_error.restype = ctypes.c_char_p
        self.user = ctypes.c_void_p()
        if self.lib.cba_rccl_create_via_file(rank, world, id_file.encode(), device, ctypes.byref(self.user)) != 0:
            raise RuntimeError("cba_rccl_create_via_file: " + self.lib.cba_rccl_last_error().decode())
        self.fn = ctypes.cast(self.lib.cba_rccl_allreduce, ctypes.c_void_p)

    def close(self):
        if self.user:
            self.lib.cba_rccl_destroy(self.user)
            self.user = ctypes.c_void_p()


# ---------------------------------------------------------------------------------------------------
# host mirror of the buffer that crosses ranks (k_pack_upper, csrc/kernels_linalg.hip): the upper 128-row blocks of
# the n_pad x n_pad reduced system -- block i keeps rows [128 i, 128 i + 128) and columns [128 i, n_pad), rows contiguous;
# the right-hand side travels in the last padding column (n_pad - 1).
# ---------------------------------------------------------------------------------------------------
def packed_upper_doubles(n_pad: int) -> int:
    return sum(128 * (n_pad - 128 * i) for i in range(n_pad // 128))


def pack_upper(S: np.ndarray) -> np.ndarray:
    n_pad = S.shape[0]
    assert S.shape == (n_pad, n_pad) and n_pad % 128 == 0
    return np.concatenate([S[128 * i:128 * i + 128, 128 * i:].ravel() for i in range(n_pad // 128)])


def unpack_upper(P: np.ndarray, n_pad: int, out: np.ndarray | None = None) -> np.ndarray:
    S = np.zeros((n_pad, n_pad)) if out is None else out
    off = 0
    for i in range(n_pad // 128):
        w = n_pad - 128 * i
        S[128 * i:128 * i + 128, 128 * i:] = P[off:off + 128 * w].reshape(128, w)
        off += 128 * w
    assert off == P.size
    return S


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


# ---------------------------------------------------------------------------------------------------------------
# Host mirror of the distributed factorisation schedule (csrc/kernels_linalg.hip: ldlt_factor_distributed), numpy.
# Same ownership rule, same three phases per panel; `exchange(buf)` sums a float64 array over the ranks in place.
# Used by the CPU tests of the multi-rank path (gloo, world size 2) -- the device kernels are covered on the GPU.
# ---------------------------------------------------------------------------------------------------------------
def distributed_ldlt_upper(S, rank: int, world: int, exchange, group: int = 512, panel: int = 512):
    """S: (n, n) symmetric positive definite, upper triangle valid (modified in place).  Every rank passes the SAME matrix
    (the all-reduced reduced system).  Returns (L, d) with S = L diag(d) L^T, complete on every rank."""
    n = S.shape[0]
    S = S.copy()
    iu = np.triu_indices(n, 1)
    S[(iu[1], iu[0])] = 0.0                                   # keep the upper triangle only, like the device storage
    owner = (np.arange(n) // group) % world
    L = np.eye(n)
    d = np.zeros(n)
    for k0 in range(0, n, panel):
        e0 = min(k0 + panel, n)
        if k0 > 0 and world > 1:                              # (1) assemble the block row from the owners of its columns
            buf = np.where(owner[None, k0:] == rank, S[k0:e0, k0:], 0.0)
            exchange(buf)
            S[k0:e0, k0:] = buf
        X = np.zeros((e0 - k0, n - k0))                       # (2) the panel, replicated: unblocked LDL^T of the block row
        for j in range(k0, e0):
            d[j] = S[j, j]
            L[j + 1:, j] = S[j, j + 1:] / d[j]                # row j of the upper storage holds column j of L (times d)
            X[j - k0, j + 1 - k0:] = S[j, j + 1:]
            # eliminate within the block row
            for i in range(j + 1, e0):
                S[i, i:] -= L[i, j] * S[j, i:]
        # (3) trailing update of the owned columns: S[m][c] -= sum_p L[m][p] d_p L[c][p], rows e0 <= m <= c
        own = np.nonzero(owner[e0:] == rank)[0] + e0
        if own.size:
            Lp = L[e0:, k0:e0]                                # (n - e0) x nb
            upd = (Lp * d[k0:e0]) @ Lp.T                      # symmetric; only the owned columns' upper part is used
            for c in own:
                S[e0:c + 1, c] -= upd[:c + 1 - e0, c - e0]
    return L, d
