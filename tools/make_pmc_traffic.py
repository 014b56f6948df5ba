#!/usr/bin/env python
"""profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE tables of tools/rocprof_pmc.py (two separate --pmc passes):
HBM bytes per launch of the dominant kernel, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950.
  python tools/make_pmc_traffic.py <fetch table> <write table> <out json> <tag>"""
import hashlib, json, os, sys


def per_call(path, counter):
    """(launches, bytes per launch, per-instantiation rows) over ALL instantiations of k_gemm_atb<128,128,...>: since round 5 the
    block-sparse Schur launch (12-row K slabs) and the dense super-panel updates (16-row) are two instantiations of the one kernel."""
    calls, kib, parts = 0, 0.0, []
    for line in open(path):
        if "k_gemm_atb<128" in line.replace(" ", "") or "k_gemm_atbILi128" in line or "k_gemm_atb<128, 128" in line:
            f = line.split()
            # ... counter calls sum_KiB per_call_MiB
            i = f.index(counter)
            calls += int(f[i + 1]); kib += float(f[i + 2])
            parts.append({"symbol": f[0][:80], "launches": int(f[i + 1]), "bytes_per_launch_raw": float(f[i + 2]) * 1024.0 / int(f[i + 1])})
    if not calls:
        raise SystemExit(f"{path}: no k_gemm_atb row")
    return calls, kib * 1024.0 / calls, parts


n_f, fetch, parts_f = per_call(sys.argv[1], "FETCH_SIZE")
n_w, write, parts_w = per_call(sys.argv[2], "WRITE_SIZE")
tag = sys.argv[4] if len(sys.argv) > 4 else "r04"
out = {"kernel": "cba::k_gemm_atb<128,128,64,64,true>", "launches": n_f,
       "fetch_size_raw_bytes_per_launch": fetch, "fetch_bytes_per_launch": 2.0 * fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": 2.0 * fetch + write,
       "instantiations_fetch_raw": parts_f, "instantiations_write": parts_w,
       "source": f"profiles/{tag}_pmc_FETCH_SIZE.txt + {tag}_pmc_WRITE_SIZE.txt: two separate `rocprofv3 --kernel-trace --pmc <counter>` passes over "
                 "`bench.py --steps 2 --warmup 0`; FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE as "
                 f"reported; average over all launches of the kernel ({n_f // 2} per step: the Schur product and the super-panel updates of the two-level factorisation)"}
# provenance: bench.py quotes this file only while kernels_linalg.hip is the source these passes ran with
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "camera_calibration_amd", "csrc", "kernels_linalg.hip")
out["kernel_source_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
