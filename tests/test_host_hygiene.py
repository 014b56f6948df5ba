"""CPU-only checks of host-side argument handling added for the round-2 review (no GPU work is launched)."""
import ctypes as C
import os
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_id_file_reader_ignores_stale_files_and_reads_fresh_ones(tmp_path):
    path = os.path.join(ROOT, "camera_calibration_amd", "libcalib_ba_rccl.so")
    if not os.path.exists(path):
        pytest.skip("libcalib_ba_rccl.so not built")
    try:
        L = C.CDLL(path)
    except OSError as ex:       # RCCL / HIP runtime not loadable on this host
        pytest.skip(str(ex))
    L.cba_rccl_debug_read_id_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    f = tmp_path / "id"
    f.write_bytes(bytes(range(128)))
    old = time.time() - 3600
    os.utime(f, (old, old))
    buf = C.create_string_buffer(128)
    assert L.cba_rccl_debug_read_id_file(str(f).encode(), buf, 50, 120) == -1          # leftover of an earlier launch
    f.write_bytes(bytes(reversed(range(128))))
    assert L.cba_rccl_debug_read_id_file(str(f).encode(), buf, 50, 120) == 0
    assert buf.raw == bytes(reversed(range(128)))
    (tmp_path / "short").write_bytes(b"x" * 5)
    assert L.cba_rccl_debug_read_id_file(str(tmp_path / "short").encode(), buf, 30, 120) == -1   # partial file


def test_distributed_solve_arguments_are_validated_before_any_device_work():
    from camera_calibration_amd import engine as eng
    L = eng.load()
    cam = eng.CbaCamera(0, 640, 480, 0, 0, 639, 479, 8, 6)
    cams = (eng.CbaCamera * 1)(cam)

    @eng.ALLREDUCE_FN
    def cb(ptr, count, user):
        return 0
    for rank, world, with_cb in ((0, 0, True), (2, 2, True), (-1, 2, True), (0, 2, False)):
        cfg = eng.CbaConfig()
        cfg.n_cameras = 1; cfg.cameras = cams; cfg.n_images = 2; cfg.n_points = 4; cfg.numerical_diff_delta = 1e-4
        cfg.distributed_solve = 1; cfg.rank = rank; cfg.world_size = world
        if with_cb:
            cfg.allreduce = cb
        out = C.c_void_p()
        assert L.cba_create(C.byref(cfg), C.byref(out)) == -1, (rank, world, with_cb)     # CBA_ERR_ARG (include/cba.h)


def test_tail_kernel_uses_m0_only_inside_its_own_lds_dma(tmp_path, monkeypatch):
    # k_ldlt_tail writes M0 from inline asm without a clobber (kernels_linalg.hip: tail_dma16); the build fails unless the ISA shows
    # that nothing else in the kernel touches M0.  Here: the check passes on the compiled object, and it catches a foreign M0 use.
    import shutil
    from camera_calibration_amd import build as hb
    obj = os.path.join(ROOT, "camera_calibration_amd", "csrc", "kernels_linalg.o")
    if not os.path.exists(obj) or shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("kernels_linalg.o / ROCm llvm tools not present")
    try:
        sites = hb.check_tail_m0(obj)
    except RuntimeError as ex:
        if "not found" in str(ex) and "llvm-objdump" in str(ex):
            pytest.skip(str(ex))
        raise
    assert sites >= 8          # 4 + 4 operand DMAs + the d vector per stage, several inlined copies
    real = hb.disassemble_device_code(obj)
    anchor = real.index("k_ldlt_tail")
    body_start = real.index("\n", anchor) + 1
    for foreign in ("\ts_mov_b32 m0, 0x100 // 000000000000: BEFC00FF\n", "\tv_readlane_b32 s4, v1, m0 // 0: 0\n", "\ts_sendmsg sendmsg(MSG_INTERRUPT) // 0: 0\n"):
        doctored = real[:body_start] + foreign + real[body_start:]
        monkeypatch.setattr(hb, "disassemble_device_code", lambda _o, d=doctored: d)
        with pytest.raises(RuntimeError, match="check_tail_m0"):
            hb.check_tail_m0(obj)
