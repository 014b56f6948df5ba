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


def stage_b(path, counter):
    """bytes per step of the accumulation stage's kernels (k_assemble, k_accumulate*, k_cell_*, k_strip_band_mask), raw counter values"""
    rows = {}
    for line in open(path):
        f = line.split()
        if counter not in f:
            continue
        name = f[0]
        if any(k in name for k in ("k_assemble", "k_accumulate", "k_cell_count", "k_cell_fill", "k_cell_scan", "k_strip_band_mask")):
            i = f.index(counter)
            rows[name[:90]] = {"launches": int(f[i + 1]), "bytes_raw": float(f[i + 2]) * 1024.0}
    return rows


n_f, fetch, parts_f = per_call(sys.argv[1], "FETCH_SIZE")
n_w, write, parts_w = per_call(sys.argv[2], "WRITE_SIZE")
tag = sys.argv[4] if len(sys.argv) > 4 else "r04"
out = {"kernel": "cba::k_gemm_atb<128,128,64,64,true>", "launches": n_f,
       "fetch_size_raw_bytes_per_launch": fetch, "fetch_bytes_per_launch": 2.0 * fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": 2.0 * fetch + write,
       "instantiations_fetch_raw": parts_f, "instantiations_write": parts_w,
       "source": f"profiles/{tag}_pmc_FETCH_SIZE.txt + {tag}_pmc_WRITE_SIZE.txt: two separate `rocprofv3 --kernel-trace --pmc <counter>` passes over "
                 "`bench.py --steps 2 --warmup 0`; FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE as "
                 f"reported; average over all launches of the kernel ({n_f // 2} per step: grid-first order = the border update; pose-first order = the Schur product and the super-panel updates)"}
# stage B (J^T J / J^T r accumulation): counters of its kernels, per step (the passes run `--steps 2 --warmup 0`: two Jacobian passes)
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
sb_f, sb_w = stage_b(sys.argv[1], "FETCH_SIZE"), stage_b(sys.argv[2], "WRITE_SIZE")
if sb_f or sb_w:
    fr = sum(r["bytes_raw"] for r in sb_f.values()) / steps
    wr = sum(r["bytes_raw"] for r in sb_w.values()) / steps
    out["stage_B"] = {"fetch_raw_bytes_per_step": fr, "write_bytes_per_step": wr,
                      "kernels_fetch": {k: {"launches_per_step": v["launches"] / steps, "bytes_raw_per_step": v["bytes_raw"] / steps} for k, v in sb_f.items()},
                      "kernels_write": {k: {"launches_per_step": v["launches"] / steps, "bytes_per_step": v["bytes_raw"] / steps} for k, v in sb_w.items()},
                      "note": "FETCH_SIZE is reported raw: MI355X_MICROARCH.md calibrates the gfx950 doubling for 16-byte-per-lane streaming reads only, and "
                              "every accumulation kernel reads its Jacobian records, masks and lists with 8-byte (or narrower) loads per lane -- no kernel of "
                              "this stage qualifies, so no correction is applied; 2 x raw is carried next to it as the upper bound (stage_rooflines.B_accumulation)"}
# provenance: bench.py quotes this file only while kernels_linalg.hip is the source these passes ran with
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "camera_calibration_amd", "csrc", "kernels_linalg.hip")
out["kernel_source_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
src_obs = os.path.join(os.path.dirname(src), "kernels_obs.hip")
out["obs_kernel_source_sha256"] = hashlib.sha256(open(src_obs, "rb").read()).hexdigest()
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
