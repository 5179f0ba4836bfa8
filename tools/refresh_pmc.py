#!/usr/bin/env python
"""Re-collect the PMC summary of the north-star kernel (backbone QKV GEMM) for the CURRENT library build.

Run on the GPU box (through gpurun), from the repo root:

    python tools/refresh_pmc.py [--out gpurun_out/pmc] [--precision fp16]

Three separate rocprofv3 --pmc passes of the bench command (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not
fit one pass; counters are never combined with the trace domains gpurun refuses):
    1. FETCH_SIZE          2. WRITE_SIZE          3. SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES
and writes <out>/qkv_gemm_pmc.json (+ the per-kernel CSV).  Copy both into profiles/ and commit them: bench.py reports
`roofline.traffic` from profiles/qkv_gemm_pmc.json ONLY while its `source_hash` equals edgecape_amd.build.source_hash() of the
library it runs - any kernel edit makes the summary stale and `traffic` null until this script has been run again.

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16-B/lane
coalesced reads (x2); WRITE_SIZE is used as reported.  The counters sit at the L2's fabric side: Infinity-Cache hits are included,
so this is fabric traffic, an upper bound on DRAM traffic.
"""
import argparse
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rocpd_pmc  # noqa: E402
from edgecape_amd import build, synth  # noqa: E402

PASSES = [("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
          ("sq", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES"]),
          ("inst", ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS"])]   # (round 5: the attention kernel's instruction mix)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc"))
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--shots", type=int, default=1)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--arch", default="dinov2_vitb14")
    args = ap.parse_args()
    args.out = os.path.abspath(args.out)          # rocprofv3 runs from /tmp
    os.makedirs(args.out, exist_ok=True)
    bench_cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-episode", "--no-alt", "--sustained-seconds", "0", "--steps", "2", "--warmup", "1",
             "--precision", args.precision, "--batch", str(args.batch), "--shots", str(args.shots), "--image-size", str(args.image_size),
             "--arch", args.arch] + (["--head-precision", "bf16x3"] if args.precision in ("bf16x3", "fp16x2") else [])
    env = dict(os.environ, TMPDIR="/tmp")
    merged = {}
    for tag, counters in PASSES:
        d = os.path.join(args.out, "pmc_" + tag)
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", d, "-o", "r", "--"] + bench_cmd
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(f"rocprofv3 pass {tag} failed:\n{r.stderr[-2000:]}")
        dbs = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)
        if not dbs:
            sys.exit(f"no rocpd database under {d}")
        for (kn, cn), (n, v, dur) in rocpd_pmc.summarise(dbs[0]).items():
            merged[(rocpd_pmc.short(kn), cn)] = (n, v / n, dur / n)
    rows = sorted(merged.items())
    sys.path.insert(0, ROOT)
    import bench
    json_name = os.path.basename(bench.pmc_path(args.batch, args.shots, args.image_size, args.arch, args.precision))
    with open(os.path.join(args.out, json_name.replace(".json", "_kernels.csv")), "w") as f:
        f.write("Kernel,Counter,Dispatches,MeanValuePerDispatch,MeanDurationNs\n")
        for (kn, cn), (n, v, dur) in rows:
            f.write(f'"{kn}",{cn},{n},{v:.3f},{dur:.1f}\n')

    f16 = "true" if args.precision == "fp16" else "false"
    x3 = args.precision == "bf16x3"    # K-concatenated form: bf16 [hi | lo | hi] x [W_hi | W_hi | W_lo], depth 3 K, fp32 output (G8_F32 kind, tag 1)
    x2 = args.precision == "fp16x2"    # fp16x2 operands: rows [fp16 | e5m2 | e5m2] x [fp16 | e4m3 | e4m3], 4 bytes per value, fp32 output (G8_F32 kind, tag 1, X2)
    name = next((kn for (kn, cn) in merged if kn.startswith("gemm8_bf16_kernel<6, 1, true, 0, true" if x2 else "gemm8_bf16_kernel<6, 1, false" if x3 else "gemm8_bf16_kernel<1, 1, " + f16)), None)
    if name is None:
        sys.exit("QKV kernel symbol not found among: " + ", ".join(sorted({k for k, _ in merged})))
    get = lambda c: merged[(name, c)][1]
    a = synth.ARCHS[args.arch]
    T = (args.image_size // 14) ** 2 + 1
    M, K, N = (1 + args.shots) * args.batch * T, a["C"], 3 * a["C"]
    algorithmic = M * K * 2 + N * K * 2 + M * N * 2 + N * 4          # A + W + C (16-bit) + bias
    if x3 or x2:
        algorithmic = M * 2 * K * 2 + N * 2 * K * 2 + M * N * 4 + N * 4   # two bf16 planes (fp16x2: one fp16 + two FP8 planes) of A and of W in memory, fp32 C
    fetch_kb, write_kb = get("FETCH_SIZE"), get("WRITE_SIZE")
    traffic = (2.0 * fetch_kb + write_kb) * 1024.0
    gui = get("GRBM_GUI_ACTIVE")
    dur_ns = merged[(name, "SQ_WAVE_CYCLES")][2]
    out = {
        "kernel": f"{name} (backbone QKV GEMM, M={M} K={3 * K if x3 else K} N={N}, {args.precision})" + (" - fp16x2: K fp16 + 2 K FP8 deep" if x2 else ""),
        "workload": [args.batch, args.shots, args.image_size, args.arch, args.precision],
        "source_hash": build.source_hash(),
        "command": "rocprofv3 --pmc <counters> --kernel-trace -- " + " ".join(bench_cmd[1:]).replace(ROOT + "/", ""),
        "passes": {t: c for t, c in PASSES},
        "dispatches": merged[(name, "FETCH_SIZE")][0],
        "fetch_size_kb_per_launch": round(fetch_kb, 3), "write_size_kb_per_launch": round(write_kb, 3), "fetch_correction": 2.0,
        "correction_note": "MI355X_MICROARCH.md §HBM: FETCH_SIZE on gfx950 reports 1/2 of the bytes of 16-B/lane coalesced reads; WRITE_SIZE as reported",
        "traffic_bytes_per_launch": round(traffic, 1), "algorithmic_bytes_per_launch": algorithmic,
        "traffic_over_algorithmic": round(traffic / algorithmic, 3),
        "mfma_busy_cycles": get("SQ_VALU_MFMA_BUSY_CYCLES"), "sq_busy_cu_cycles": get("SQ_BUSY_CU_CYCLES"), "gui_active_cycles_sum": gui,
        # MFMA busy cycles per CU-cycle of the launch: busy cycles are summed over the 1024 SIMDs' pipes, GRBM_GUI_ACTIVE over the 8 XCDs
        "mfma_util": round(get("SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8.0 * 256.0 * 4.0), 4) if gui > 0 else None,
        "lds_bank_conflict_cycles": get("SQ_LDS_BANK_CONFLICT"), "sq_wave_cycles": get("SQ_WAVE_CYCLES"),
        "mean_duration_us_profiled": round(dur_ns / 1e3, 2),
        "effective_clock_ghz": round(gui / 8.0 / dur_ns, 3) if dur_ns > 0 else None,
    }
    with open(os.path.join(args.out, json_name), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
