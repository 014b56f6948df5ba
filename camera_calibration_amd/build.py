"""Builds libcalib_ba_hip.so (hand-written HIP for gfx950) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcalib_ba_hip.so")
SOURCES = ["cba_api.hip", "kernels_obs.hip", "kernels_linalg.hip", "kernels_fit.hip", "gridfirst_plan.hip", "kernels_gridfirst.hip"]
HEADERS = ["cba_internal.h", "model.hip.h", "gridfirst_plan.h", os.path.join("..", "..", "include", "cba.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]  # keep MFMA accumulators in VGPRs: no AGPR<->VGPR copies in the K loop


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


FLAGS_STAMP = os.path.join(HERE, ".build_flags")      # the extra flags the current library was built with ("" = the product build)


def _extra_flags() -> str:
    return " ".join(os.environ.get("CBA_BUILD_EXTRA_FLAGS", "").split())


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    built_with = open(FLAGS_STAMP).read() if os.path.exists(FLAGS_STAMP) else ""
    if built_with != _extra_flags():            # a library tuned by a leftover CBA_BUILD_EXTRA_FLAGS is never reused silently
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    extra = os.environ.get("CBA_BUILD_EXTRA_FLAGS", "").split()      # developer A/B builds only (e.g. -DCBA_FD_POOL_WAVES_CENTRAL=3)
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *extra, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed for {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    check_tail_m0(os.path.join(CSRC, "kernels_linalg.o"))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    with open(FLAGS_STAMP, "w") as f:
        f.write(_extra_flags())
    return LIB


def _llvm_tool(name: str) -> str:
    for d in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin"), "/opt/rocm/llvm/bin"):
        cand = os.path.join(d, name)
        if os.path.exists(cand):
            return cand
    raise RuntimeError(name + " not found")


def disassemble_device_code(obj: str) -> str:
    """gfx950 ISA of a compiled .hip object (llvm-objdump --offloading unpacks next to its input: done in a scratch directory)."""
    import tempfile
    objdump = _llvm_tool("llvm-objdump")
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "k.o")
        shutil.copy(obj, local)
        subprocess.check_call([objdump, "--offloading", local], stdout=subprocess.DEVNULL, cwd=tmp)
        dev = [f for f in os.listdir(tmp) if "gfx950" in f]
        if len(dev) != 1:
            raise RuntimeError(f"expected one gfx950 code object in {obj}, found {dev}")
        return subprocess.check_output([objdump, "-d", os.path.join(tmp, dev[0])], text=True)


# the helper's own write of M0: the "s" operand of the inline asm is any scalar source the compiler picks (an SGPR, vcc_lo / vcc_hi, a
# trap temporary, a literal)
_M0_MOV = r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+|0x[0-9a-f]+|\d+)"
# instructions that read M0 without naming it as an operand
_IMPLICIT_M0 = ("s_sendmsg", "s_movrel", "v_movrel", "ds_gws", "s_ttrace", "v_interp", "ds_ordered_count")


def check_tail_m0(obj: str) -> int:
    """Build-time guard for the inline-asm LDS-DMA of k_ldlt_tail (kernels_linalg.hip: tail_dma16 / tail_dma4).

    Those helpers write M0 (the LDS base of global_load_lds) from inline asm and cannot declare it clobbered (the compiler rejects
    the clobber).  That is only correct while the compiler itself never keeps a value in M0 inside that kernel.  This check
    disassembles the kernel and fails the build unless EVERY M0 access in it is one of ours: `s_mov_b32 m0, sN`, `s_nop 0`,
    `global_load_lds_dword[x4]`, in that order, and no instruction with an implicit M0 operand appears.  Returns the number of
    DMA sites checked."""
    import re
    asm = disassemble_device_code(obj)
    sites = 0
    for kernel in TAIL_KERNELS:
        sites += _check_kernel_m0(asm, kernel, obj)
    return sites


# every kernel that inlines tail_dma16 / tail_dma4 (the dense dataflow launch and the block-sparse one of the grid-first order)
TAIL_KERNELS = ("k_ldlt_tail", "k_ldlt_sparse")


def _check_kernel_m0(asm: str, kernel: str, obj: str) -> int:
    import re
    m = re.search(r"^[0-9a-f]+ <[^>]*" + kernel + r"[^>]*>:\n(.*?)(?=^[0-9a-f]+ <[^>]*>:|\Z)", asm, flags=re.S | re.M)
    if not m:
        raise RuntimeError("check_tail_m0: " + kernel + " not found in " + obj)
    ins = [ln.split("//")[0].strip() for ln in m.group(1).splitlines() if ln.strip()]
    sites = 0
    for i, text in enumerate(ins):
        mnem = text.split()[0] if text else ""
        if any(mnem.startswith(x) for x in _IMPLICIT_M0):
            raise RuntimeError(f"check_tail_m0: {kernel} contains `{text}` (implicit M0 operand) next to the inline-asm LDS-DMA")
        if mnem.startswith("global_load_lds") or (mnem.startswith("buffer_load") and " lds" in text):
            if i < 2 or not re.fullmatch(_M0_MOV, ins[i - 2]) or ins[i - 1] != "s_nop 0":
                raise RuntimeError(f"check_tail_m0: LDS-DMA `{text}` in {kernel} is not preceded by the helper's own `s_mov_b32 m0` / `s_nop 0`")
            sites += 1
        elif re.search(r"\bm0\b", text):
            ok = re.fullmatch(_M0_MOV, text) and i + 2 < len(ins) and ins[i + 1] == "s_nop 0" and ins[i + 2].startswith("global_load_lds")
            if not ok:
                raise RuntimeError(f"check_tail_m0: {kernel} uses M0 outside tail_dma16 / tail_dma4: `{text}` -- the inline asm there "
                                   "writes M0 without a clobber; route that use around M0 or move the DMA to the builtin")
    if sites == 0:
        raise RuntimeError(f"check_tail_m0: no LDS-DMA found in {kernel} (did the helper loop change? update this check)")
    return sites


HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libcalib_ba_host.so")
HOST_SOURCES = ["joint_optimization_hip.cc", "calibration_report_hip.cc", "calibration_io.cc", "central_generic_fit_hip.cc", "calibration_hip.cc"]
# test scaffolding (extern "C" entry points that build Dataset / BAState objects from packed arrays for the Python tests):
# its own library, NOT part of the product library
HOST_TEST_LIB = os.path.join(HERE, "libcalib_ba_host_test.so")
HOST_TEST_SOURCES = ["host_test_shim.cc"]
HOST_HEADERS = ["vis_types.h", "camera_model.h", "dataset.h", "joint_optimization.h"]


def build_host(force: bool = False) -> str:
    """C++ host adapter (reference API mirror) linked against libcalib_ba_hip.so."""
    deps = [os.path.join(HOST_DIR, f) for f in HOST_SOURCES + HOST_HEADERS] + [LIB]
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB,
           *[os.path.join(HOST_DIR, f) for f in HOST_SOURCES], "-L" + HERE, "-lcalib_ba_hip", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return HOST_LIB


RCCL_LIB = os.path.join(HERE, "libcalib_ba_rccl.so")


def build_rccl(force: bool = False) -> str:
    """Native RCCL all-reduce callback for C++ hosts (include/cba_rccl.h); links librccl, the engine does not."""
    src = os.path.join(HOST_DIR, "rccl_allreduce.cc")
    hdr = os.path.join(HERE, "..", "include", "cba_rccl.h")
    if not force and os.path.exists(RCCL_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(RCCL_LIB) for d in (src, hdr)):
        return RCCL_LIB
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-x", "hip", "--offload-arch=gfx950", src, "-o", RCCL_LIB,
           "-I" + os.path.join(rocm, "include"), "-L" + os.path.join(rocm, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.check_call(cmd)
    return RCCL_LIB


def build_host_test(force: bool = False) -> str:
    """Test-only shim over the C++ host adapter (tests/test_gpu_host_adapter.py etc.)."""
    build_host(force)
    deps = [os.path.join(HOST_DIR, f) for f in HOST_TEST_SOURCES + HOST_HEADERS] + [HOST_LIB]
    if not force and os.path.exists(HOST_TEST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_TEST_LIB) for d in deps):
        return HOST_TEST_LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_TEST_LIB,
           *[os.path.join(HOST_DIR, f) for f in HOST_TEST_SOURCES], "-L" + HERE, "-lcalib_ba_host", "-lcalib_ba_hip", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return HOST_TEST_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv))
    print(build_host_test(force="--force" in sys.argv))
    print(build_rccl(force="--force" in sys.argv))
