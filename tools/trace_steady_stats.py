"""Per-kernel statistics of the STEADY-STATE optimiser steps of a rocprofv3 --kernel-trace run of bench.py.

    python tools/trace_steady_stats.py <..._kernel_trace.csv> <out.csv> [steps]

rocprofv3's own --stats table sums the whole process: model initialisation, the warm-up steps and the one-off launches of
bench.py's in-kernel clock probe (2 s of back-to-back GEMMs) and of the grouped-GEMM autotune -- thousands of extra GEMM
launches that belong to no step.  This tool cuts the trace at the optimiser launches (one
`adamw*_multi_kernel` per step; the last launch of a step) and keeps the LAST `steps` steps.  Columns as rocprofv3's
kernel_stats.csv plus CallsPerStep / MsPerStep."""
import csv, sys


def main():
    path, out = sys.argv[1:3]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if "adamw" in r["Kernel_Name"] and "multi_kernel" in r["Kernel_Name"]]
    assert len(ends) > steps, f"only {len(ends)} optimiser launches in the trace"
    t0, t1 = ends[-steps - 1], ends[-1]
    agg = {}
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > t0 and e <= t1:
            a = agg.setdefault(r["Kernel_Name"], [])
            a.append(e - s)
    total = sum(sum(v) for v in agg.values())
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsPerStep", "MsPerStep"])
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), f"{sum(v)/len(v):.1f}", f"{100.0*sum(v)/total:.2f}", min(v), max(v), f"{len(v)/steps:.1f}",
                        f"{sum(v)/steps/1e6:.3f}"])
    print(f"{steps} steady steps, wall {(t1 - t0)/steps/1e6:.2f} ms/step, kernel time {total/steps/1e6:.2f} ms/step, {len(agg)} kernels -> {out}")


if __name__ == "__main__":
    main()
