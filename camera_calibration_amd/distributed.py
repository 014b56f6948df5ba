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


class GridFirstSharedLayout:
    """FOLLOW-UP DESIGN (not used by the engine yet): what would cross the ranks per Gauss-Newton step if the
    image-sharded path used the grid-first elimination order -- literally BASELINE.json's "shared intrinsics and pattern J^T J /
    J^T r blocks" instead of the packed reduced system S of the pose-first order:

        [ grid x grid, BANDED: unknown g of a camera (numbered along the short grid dimension) x its hb + 1 right neighbours |
          rig / point rows x grid columns, dense | rig rows x (rig, point) columns | 3 x 3 point blocks (upper) | J^T r of all of them ]

    An observation touches a 4 x 4 window of control points (APP/models/central_grid.h:199-209), so nothing else of the dense part of
    the reference's accumulator (LV/lm_optimizer_update_accumulator.h:108-155: rig | points | intrinsics) can be non-zero.  At BASELINE
    configs[1] this is 228 MB against the 642 MB of the packed upper triangle of S (cba_reduce_buffer_doubles).  The pose blocks D_i /
    strips B_i stay on their owners in both designs.  pack / unpack work on the reference-order dense part (oracle.System.dense_H /
    dense_b layout); tests/test_distributed_gloo.py sums the buffer over two ranks with gloo and compares with the single-process
    accumulator."""

    def __init__(self, cameras, n_points: int):
        self.cameras = list(cameras)
        C = len(self.cameras)
        self.rig = 6 * C if C > 1 else 0
        self.n_rp = self.rig + 3 * n_points
        self.n_points = n_points
        self.cam_first, self.hb, self.order = [], [], []
        off = self.n_rp
        for cam in self.cameras:
            ppg = cam.params_per_grid_point
            gw, gh = cam.grid_w, cam.grid_h
            long_is_x = gw >= gh
            nl, ns = (gw, gh) if long_is_x else (gh, gw)
            # band order: along the short dimension inside a line of the long one; dense column of every unknown in that order
            cols = np.empty(ppg * gw * gh, dtype=np.int64)
            k = 0
            for l in range(nl):
                for t in range(ns):
                    gx, gy = (l, t) if long_is_x else (t, l)
                    for d in range(ppg):
                        cols[k] = off + ppg * (gx + gy * gw) + d
                        k += 1
            self.order.append(cols)
            self.hb.append((3 * ns + 3) * ppg + ppg - 1)
            self.cam_first.append(off)
            off += ppg * gw * gh
        self.dense_dof = off
        self.G = off - self.n_rp
        # offsets of the parts
        self.off_band = 0
        n = 0
        self.band_rows = []
        for cols, hb in zip(self.order, self.hb):
            self.band_rows.append((n, cols.size, hb + 1))
            n += cols.size * (hb + 1)
        self.off_rp_grid = n
        n += self.n_rp * self.G
        self.off_rig = n
        n += self.rig * self.n_rp
        self.off_pp = n
        n += 6 * n_points
        self.off_b = n
        n += self.dense_dof
        self.doubles = n

    def pack(self, dense_H: np.ndarray, dense_b: np.ndarray) -> np.ndarray:
        """dense_H: upper triangle in the reference order (rig | points | grids row-major)."""
        H = np.triu(dense_H) + np.triu(dense_H, 1).T
        buf = np.zeros(self.doubles)
        for (o, n, w), cols in zip(self.band_rows, self.order):
            band = np.zeros((n, w))
            for k in range(w):
                band[:n - k, k] = H[cols[:n - k], cols[k:]]
            buf[o:o + n * w] = band.ravel()
        gcols = np.concatenate(self.order)
        buf[self.off_rp_grid:self.off_rp_grid + self.n_rp * self.G] = H[:self.n_rp][:, gcols].ravel()
        if self.rig:
            buf[self.off_rig:self.off_rig + self.rig * self.n_rp] = H[:self.rig, :self.n_rp].ravel()
        iu = np.triu_indices(3)
        for p in range(self.n_points):
            a = self.rig + 3 * p
            buf[self.off_pp + 6 * p:self.off_pp + 6 * p + 6] = H[a:a + 3, a:a + 3][iu]
        buf[self.off_b:self.off_b + self.dense_dof] = dense_b
        return buf

    def unpack(self, buf: np.ndarray):
        """(dense_H upper triangle, dense_b) in the reference order."""
        D = self.dense_dof
        H = np.zeros((D, D))                      # filled symmetrically by ASSIGNMENT (a band pair can sit on either side of the diagonal)
        for (o, n, w), cols in zip(self.band_rows, self.order):
            band = buf[o:o + n * w].reshape(n, w)
            for k in range(w):
                H[cols[:n - k], cols[k:]] = band[:n - k, k]
                H[cols[k:], cols[:n - k]] = band[:n - k, k]
        gcols = np.concatenate(self.order)
        blk = buf[self.off_rp_grid:self.off_rp_grid + self.n_rp * self.G].reshape(self.n_rp, self.G)
        H[np.ix_(np.arange(self.n_rp), gcols)] = blk
        H[np.ix_(gcols, np.arange(self.n_rp))] = blk.T
        iu = np.triu_indices(3)
        for p in range(self.n_points):
            a = self.rig + 3 * p
            m = np.zeros((3, 3))
            m[iu] = buf[self.off_pp + 6 * p:self.off_pp + 6 * p + 6]
            H[a:a + 3, a:a + 3] = m + np.triu(m, 1).T
        if self.rig:
            r = buf[self.off_rig:self.off_rig + self.rig * self.n_rp].reshape(self.rig, self.n_rp)
            H[:self.rig, :self.n_rp] = r
            H[:self.n_rp, :self.rig] = r.T
        return np.triu(H), buf[self.off_b:self.off_b + D].copy()


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
        self.collective_fn = ctypes.cast(self.lib.cba_rccl_collective, ctypes.c_void_p)      # cba_config.collective

    def comm_count(self) -> int:
        """ncclCommCount of the communicator (what bench.py reports as rccl_ranks on the native path)."""
        self.lib.cba_rccl_comm_count.argtypes = [ctypes.c_void_p]
        return int(self.lib.cba_rccl_comm_count(self.user))

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
# cba_collective_fn for the distributed reduced solve: reduce-scatter / all-gather on raw device pointers.
# ---------------------------------------------------------------------------------------------------------------
COLL_ALLREDUCE_SUM, COLL_REDUCE_SCATTER_SUM, COLL_ALLGATHER = 0, 1, 2


class _DeviceView:
    """A raw device pointer as a 1-D fp64 array for torch (zero copy, __cuda_array_interface__ v2)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def make_collective(local_rank: int):
    """collective(op, send_ptr, recv_ptr, count) -> 0 with torch.distributed (backend nccl = RCCL over xGMI): ncclAllReduce /
    ncclReduceScatter / ncclAllGather straight on the engine's staging buffers; waits on torch's current stream only."""
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", local_rank)

    def view(ptr, count):
        return torch.as_tensor(_DeviceView(ptr, count), device=dev)

    def collective(op: int, send: int, recv: int, count: int) -> int:
        world = dist.get_world_size()
        if op == COLL_ALLREDUCE_SUM:
            dist.all_reduce(view(recv, count))
        elif op == COLL_REDUCE_SCATTER_SUM:
            dist.reduce_scatter_tensor(view(recv, count), view(send, count * world))
        elif op == COLL_ALLGATHER:
            dist.all_gather_into_tensor(view(recv, count * world), view(send, count))
        else:
            return 1
        torch.cuda.current_stream().synchronize()
        return 0

    return collective


def make_collective_host_staged():
    """The same three collectives through HOST memory with whatever process group is initialised (gloo): several ranks on ONE
    GPU in the tests (RCCL refuses two ranks on one device)."""
    import torch
    import torch.distributed as dist

    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemcpy.restype = ctypes.c_int
    D2H, H2D = 2, 1

    def pull(ptr, count):
        host = np.empty(count, dtype=np.float64)
        if hip.hipMemcpy(host.ctypes.data, ptr, count * 8, D2H) != 0:
            raise RuntimeError("hipMemcpy D2H")
        return host

    def push(ptr, host):
        if hip.hipMemcpy(ptr, host.ctypes.data, host.size * 8, H2D) != 0:
            raise RuntimeError("hipMemcpy H2D")

    def collective(op: int, send: int, recv: int, count: int) -> int:
        world, rank = dist.get_world_size(), dist.get_rank()
        if op == COLL_ALLREDUCE_SUM:
            h = pull(recv, count)
            dist.all_reduce(torch.from_numpy(h))
            push(recv, h)
        elif op == COLL_REDUCE_SCATTER_SUM:          # gloo has no reduce-scatter: sum everything, keep the own block
            h = pull(send, count * world)
            dist.all_reduce(torch.from_numpy(h))
            push(recv, np.ascontiguousarray(h[rank * count:(rank + 1) * count]))
        elif op == COLL_ALLGATHER:
            h = pull(send, count)
            out = [torch.empty(count, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(out, torch.from_numpy(h))
            push(recv, np.concatenate([t.numpy() for t in out]))
        else:
            return 1
        return 0

    return collective


# ---------------------------------------------------------------------------------------------------------------
# Host mirror of the distributed factorisation schedule (csrc/kernels_linalg.hip: ldlt_factor_distributed), numpy: same
# ownership rule, same transfers (dist_rect: which rows of which column groups travel, and where they sit in the
# buffers), same order of collectives.  `collective(op, send, recv)` works on float64 arrays (recv in place), with
# the semantics of cba_collective_fn.  Used by the CPU tests of the multi-rank path (gloo, world size 2 and 3) with the
# schedule scaled down (group / W / tail_rows are parameters) -- the device kernels are covered on the GPU.
# ---------------------------------------------------------------------------------------------------------------
def dist_rect(n_pad: int, group: int, g_begin: int, world: int, R0: int, nrows: int, q: int, i: int):
    """i-th column group of rank q in a transfer: (col0, width, height, offset in q's block) or None (kernels_linalg.hip: dist_rect)."""
    gq0 = g_begin + (q - g_begin % world) % world
    g = gq0 + i * world
    col0 = g * group
    if col0 >= n_pad:
        return None
    width = min(group, n_pad - col0)
    if nrows > 0:
        return col0, width, nrows, i * nrows * group
    h0 = (gq0 + 1) * group - R0
    height = min(col0 + group, n_pad) - R0
    return col0, width, height, group * (i * h0 + world * group * (i * (i - 1) // 2))


def _dist_rects(n_pad, group, g_begin, world, R0, nrows, q):
    out, i = [], 0
    while True:
        r = dist_rect(n_pad, group, g_begin, world, R0, nrows, q, i)
        if r is None:
            return out
        out.append(r)
        i += 1


def _dist_count(n_pad, group, g_begin, world, R0, nrows):
    count = 0
    for q in range(world):
        for col0, width, height, off in _dist_rects(n_pad, group, g_begin, world, R0, nrows, q):
            count = max(count, off + height * width)
    return count


def _dist_copy(S, buf, n_pad, group, g_begin, world, R0, nrows, q, unpack):
    for col0, width, height, off in _dist_rects(n_pad, group, g_begin, world, R0, nrows, q):
        blk = buf[off:off + height * width].reshape(height, width)
        if unpack:
            S[R0:R0 + height, col0:col0 + width] = blk
        else:
            blk[:, :] = S[R0:R0 + height, col0:col0 + width]


def distributed_ldlt_upper(S_partial, rank: int, world: int, collective, group: int = 512, W: int = 2048, tail_rows: int = 6144):
    """S_partial: (n, n), this rank's PARTIAL sum of the symmetric positive definite reduced system (upper triangle valid;
    n a multiple of `group`'s tile, W a multiple of `group`).  Returns (L, d) of the SUM over ranks, S = L diag(d) L^T,
    complete and identical on every rank."""
    n = S_partial.shape[0]
    S = np.triu(S_partial).copy()                       # keep the upper triangle only, like the device storage
    L = np.eye(n)
    d = np.zeros(n)
    owner = (np.arange(n) // group) % world

    def factor_rows(k0, e0):                            # the dataflow launch: rows [k0, e0) with their whole row strip
        for j in range(k0, e0):
            d[j] = S[j, j]
            L[j + 1:, j] = S[j, j + 1:] / d[j]
            for i in range(j + 1, e0):
                S[i, i:] -= L[i, j] * S[j, i:]

    def update_owned(k0, e0, r_begin, r_end):           # rows [r_begin, r_end) of the owned columns -= L^T D L with K = [k0, e0)
        if r_end <= r_begin:
            return
        Lr = L[r_begin:r_end, k0:e0] * d[k0:e0]
        for c in np.nonzero(owner == rank)[0]:
            if c < r_begin:
                continue
            hi = min(c + 1, r_end)
            S[r_begin:hi, c] -= Lr[:hi - r_begin] @ L[c, k0:e0]

    nsp, k0 = 0, 0
    while n - k0 > tail_rows + W // 2 and n - (k0 + W) >= W // 2:
        nsp += 1
        k0 += W
    if nsp == 0:
        collective(COLL_ALLREDUCE_SUM, None, S.reshape(-1))
        factor_rows(0, n)
        return L, d
    # (1) first band summed everywhere, the rest reduce-scattered into the owners
    band = S[:W].reshape(-1)
    collective(COLL_ALLREDUCE_SUM, None, band)
    count = _dist_count(n, group, W // group, world, W, 0)
    send = np.zeros(world * count)
    recv = np.zeros(count)
    for q in range(world):
        _dist_copy(S, send[q * count:(q + 1) * count], n, group, W // group, world, W, 0, q, unpack=False)
    collective(COLL_REDUCE_SCATTER_SUM, send, recv)
    _dist_copy(S, recv, n, group, W // group, world, W, 0, rank, unpack=True)
    # (2) super-panels
    for k in range(nsp):
        k0, e0 = k * W, k * W + W
        last = k == nsp - 1
        e1 = n if last else e0 + W
        factor_rows(k0, e0)
        update_owned(k0, e0, e0, e1)
        if not last:
            update_owned(k0, e0, e1, n)
        count = _dist_count(n, group, e0 // group, world, e0, e1 - e0)
        send = np.zeros(count)
        recv = np.zeros(world * count)
        _dist_copy(S, send, n, group, e0 // group, world, e0, e1 - e0, rank, unpack=False)
        collective(COLL_ALLGATHER, send, recv)
        for q in range(world):
            _dist_copy(S, recv[q * count:(q + 1) * count], n, group, e0 // group, world, e0, e1 - e0, q, unpack=True)
    # (3) the rest, replicated
    factor_rows(nsp * W, n)
    return L, d
