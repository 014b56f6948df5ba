"""integration/reference.patch -- the reference-side edit of INTEGRATION.md section 2 -- applies cleanly to the reference tree."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "reference.patch")
REFERENCE = "/root/reference"
EXPECTED = [
    "applications/camera_calibration/CMakeLists.txt",
    "applications/camera_calibration/src/camera_calibration/bundle_adjustment/joint_optimization.cc",
    "applications/camera_calibration/src/camera_calibration/bundle_adjustment/joint_optimization.h",
    "applications/camera_calibration/src/camera_calibration/bundle_adjustment/joint_optimization_hip.cc",
    "applications/camera_calibration/src/camera_calibration/main.cc",
    "applications/camera_calibration/src/camera_calibration/models/camera_model.h",
    "applications/camera_calibration/src/camera_calibration/models/central_generic.h",
    "applications/camera_calibration/src/camera_calibration/models/noncentral_generic.h",
]


def test_patch_touches_exactly_the_documented_files():
    with open(PATCH, encoding="utf-8") as f:
        files = sorted({line.split()[1][2:] for line in f if line.startswith("+++ b/")})
    assert files == sorted(EXPECTED)


def test_adapter_in_the_patch_uses_only_declared_c_abi_symbols():
    import re
    with open(PATCH, encoding="utf-8") as f:
        used = set(re.findall(r"\b(cba_[a-z_]+)\s*\(", f.read()))
    with open(os.path.join(ROOT, "include", "cba.h"), encoding="utf-8") as f:
        declared = set(re.findall(r"\b(cba_[a-z_]+)\s*\(", f.read()))
    assert used and used <= declared | {"cba_config"}, used - declared


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "applications")) or shutil.which("patch") is None,
                    reason="needs /root/reference and patch(1)")
def test_patch_applies_cleanly_to_the_reference_tree():
    r = subprocess.run(["patch", "-p1", "--dry-run", "--batch", "-d", REFERENCE, "-i", PATCH], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout
    assert r.stdout.count("checking file") == len(EXPECTED)
