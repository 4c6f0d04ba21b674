"""rocprofv3 SQ counter CSV of tools/pmc_gemm.py -> profiles/<tag>_gemm_pmc_sq.{csv,json}.

    python tools/pmc_postprocess_sq.py <counter_collection.csv> <tag> [reps]

Keeps the gemm_256_p8 rows (trimmed copy of the CSV) and reports the LAST of the `reps` launches of each shape (launch order
= tools/pmc_gemm.py's loop order)."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_kernels import SHAPES_3B  # noqa: E402

COUNTERS = "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAVES"


def main():
    path, tag = sys.argv[1:3]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows = [r for r in csv.DictReader(open(path)) if "gemm_256_p8" in r["Kernel_Name"]]
    by_disp = {}
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = r
    disp = sorted(by_disp)
    sites, i = {}, 0
    for name, (M, N, K) in SHAPES_3B.items():
        for kind, (m, n, k) in (("fprop", (M, N, K)), ("dgrad", (M, K, N)), ("wgrad", (N, K, M))):
            d = by_disp[disp[(i + 1) * reps - 1]]
            i += 1
            any_row = next(iter(d.values()))
            us = (int(any_row["End_Timestamp"]) - int(any_row["Start_Timestamp"])) / 1e3
            v = {c: float(d[c]["Counter_Value"]) for c in d}
            n_mfma = (m // 16) * (n // 16) * (k // 128)
            wc = v["SQ_WAVE_CYCLES"]
            sites[f"{m}x{n}x{k}"] = {
                "site": f"{name} {kind}", "us": us, "tflops": 2.0 * m * n * k / us / 1e6,
                "SQ_VALU_MFMA_BUSY_CYCLES": v["SQ_VALU_MFMA_BUSY_CYCLES"],
                "cycles_per_mfma": v["SQ_VALU_MFMA_BUSY_CYCLES"] / n_mfma,
                "mfma_busy_per_simd": v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024,
                "mfma_util_at_2.4GHz": v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (us * 2400.0),
                "wave_cycles_share": {"wait_any": v["SQ_WAIT_ANY"] / wc, "wait_inst_any": v["SQ_WAIT_INST_ANY"] / wc,
                                      "active_inst_any": v["SQ_ACTIVE_INST_ANY"] / wc},
                "SQ_LDS_BANK_CONFLICT": v["SQ_LDS_BANK_CONFLICT"]}
    out_csv = os.path.join(ROOT, "profiles", f"{tag}_gemm_pmc_sq.csv")
    with open(out_csv, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    out = os.path.join(ROOT, "profiles", f"{tag}_gemm_pmc_sq.json")
    json.dump({"source": f"rocprofv3 --pmc {COUNTERS} --kernel-trace -- python3 tools/pmc_gemm.py {reps} (last launch of each shape)",
               "note": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (duration x 2.4 GHz) = fraction of the dense FP8 peak (32 cycles per "
                       "16x16x128 MFMA); the chip runs at 1.6-1.8 GHz under this load (roofline.clock_ghz of bench.py), so the matrix "
                       "pipes are busy a larger share of the ACTUAL cycles", "sites": sites}, open(out, "w"), indent=1)
    for k, s in sites.items():
        print(f"{s['site']:12s} {k:18s} {s['us']:8.1f} us {s['tflops']:7.0f} TF  mfma util@2.4GHz {s['mfma_util_at_2.4GHz']:.3f}  "
              f"wait {s['wave_cycles_share']['wait_any']:.2f} stall {s['wave_cycles_share']['wait_inst_any']:.2f} issue {s['wave_cycles_share']['active_inst_any']:.2f}")
    print("wrote", out)


if __name__ == "__main__":
    main()
