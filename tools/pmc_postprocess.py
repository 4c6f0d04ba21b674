"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs of tools/pmc_gemm.py -> profiles/<tag>_gemm_pmc_traffic.json.

    python tools/pmc_postprocess.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag>

Keeps the gemm_256_p8 rows only (trimmed CSV copies go to profiles/ too) and takes the LAST of the `reps` launches of each
shape (launch order = tools/pmc_gemm.py's loop order).  Corrections as MI355X_MICROARCH.md 'HBM' prescribes: FETCH_SIZE is
in KiB and tallies 128-B requests at 64 B on gfx950 (x2); WRITE_SIZE in KiB, exact."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_kernels import SHAPES_3B  # noqa: E402


def rows_of(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if "gemm_256_p8" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            out.append(r)
    out.sort(key=lambda r: int(r["Dispatch_Id"]))
    return out


def main():
    fetch_csv, write_csv, tag = sys.argv[1:4]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    f, w = rows_of(fetch_csv, "FETCH_SIZE"), rows_of(write_csv, "WRITE_SIZE")
    sites = []
    for name, (M, N, K) in SHAPES_3B.items():
        for kind, (m, n, k) in (("fprop", (M, N, K)), ("dgrad", (M, K, N)), ("wgrad", (N, K, M))):
            sites.append((f"{name} {kind}", m, n, k))
    assert len(f) == len(w) == len(sites) * reps, (len(f), len(w), len(sites) * reps)
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/pmc_gemm.py {reps}; last launch of each shape",
           "corrections": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md 'HBM'); WRITE_SIZE (KiB) x 1024 exact",
           "note": "TCC_EA fabric requests: Infinity-Cache (MALL) hits are counted, so this is L2-miss traffic, an upper bound on HBM bytes.",
           "kernel": "gemm_256_p8 (algo 0 -> 4, tile shape by pick_tile_cfg)", "sites": {}}
    for i, (site, m, n, k) in enumerate(sites):
        fr, wr = f[i * reps + reps - 1], w[i * reps + reps - 1]
        fk, wk = float(fr["Counter_Value"]), float(wr["Counter_Value"])
        out["sites"][f"{m}x{n}x{k}"] = {"site": site, "kernel": fr["Kernel_Name"], "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                                        "fabric_read_bytes": fk * 1024 * 2, "fabric_write_bytes": wk * 1024,
                                        "algorithmic_bytes": m * k + n * k + 2 * m * n}
    prof = os.path.join(ROOT, "profiles")
    json.dump(out, open(os.path.join(prof, f"{tag}_gemm_pmc_traffic.json"), "w"), indent=1)
    for src, name, rows in ((fetch_csv, "fetch_size", f), (write_csv, "write_size", w)):
        with open(os.path.join(prof, f"{tag}_gemm_pmc_{name}.csv"), "w", newline="") as fh:
            wr_ = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            wr_.writeheader()
            wr_.writerows(rows)
    for k_, v in out["sites"].items():
        print(k_, v["site"], f"read x{v['fabric_read_bytes'] / (v['algorithmic_bytes'] - v['fabric_write_bytes']):.2f} of operand bytes, write x{v['fabric_write_bytes'] / (2 * int(k_.split('x')[0]) * int(k_.split('x')[1])):.2f}")


if __name__ == "__main__":
    main()
