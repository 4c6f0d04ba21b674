"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs of tools/pmc_hbm_kernels.py -> profiles/<tag>_hbm_pmc_traffic.json.

    python tools/pmc_postprocess_hbm.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag> [reps]

Per kernel family (launch order of tools/pmc_hbm_kernels.py: cast_amax x reps, swiglu fwd x reps, dswiglu x reps, mxfp8_quant
x reps) the LAST launch is reported.  Corrections as MI355X_MICROARCH.md 'HBM' prescribes: FETCH_SIZE is in KiB and tallies
128-B requests at 64 B on gfx950 (x2); WRITE_SIZE in KiB, exact."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, H, F = 8192, 3072, 8192
# algorithmic bytes per launch: reads, writes (bf16 in, fp8 out + transposed copy; MXFP8 adds 1 scale byte per 32)
FAMILIES = [
    ("cast_amax", "cast_amax_kernel", M * H * 2, M * H * 2),
    ("swiglu_cast", "swiglu_cast_kernel", M * 2 * F * 2, M * F * 2),
    ("dswiglu_cast", "swiglu_cast_kernel", M * 2 * F * 2 + M * F * 2, M * 2 * F * 2),
    ("mxfp8_quant", "mxfp8_quant_kernel", M * H * 2, M * H * 2 + 2 * M * H // 32),
]


def rows_of(path, counter):
    out = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and
           any(k in r["Kernel_Name"] for k in ("cast_amax_kernel", "swiglu_cast_kernel", "mxfp8_quant_kernel"))]
    out.sort(key=lambda r: int(r["Dispatch_Id"]))
    return out


def main():
    fetch_csv, write_csv, tag = sys.argv[1:4]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    f, w = rows_of(fetch_csv, "FETCH_SIZE"), rows_of(write_csv, "WRITE_SIZE")
    assert len(f) == len(w) == reps * len(FAMILIES), (len(f), len(w))
    res = []
    for i, (name, kern, rd, wr) in enumerate(FAMILIES):
        fr, wrw = f[(i + 1) * reps - 1], w[(i + 1) * reps - 1]
        assert kern in fr["Kernel_Name"] and kern in wrw["Kernel_Name"], (name, fr["Kernel_Name"])
        fetch = float(fr["Counter_Value"]) * 1024 * 2
        write = float(wrw["Counter_Value"]) * 1024
        res.append({"kernel": name, "shape": f"{M}x{H if 'swiglu' not in name else F}", "algorithmic_read_bytes": rd,
                    "algorithmic_write_bytes": wr, "fetch_bytes": fetch, "write_bytes": write,
                    "fetch_over_algorithmic": fetch / rd, "write_over_algorithmic": write / wr})
    out = os.path.join(ROOT, "profiles", f"{tag}_hbm_pmc_traffic.json")
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 tools/pmc_hbm_kernels.py",
               "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE KiB x 1024", "kernels": res},
              open(out, "w"), indent=1)
    for r in res:
        print(f"{r['kernel']:14s} fetch {r['fetch_bytes']/1e6:8.1f} MB = {r['fetch_over_algorithmic']:.2f}x algorithmic   "
              f"write {r['write_bytes']/1e6:8.1f} MB = {r['write_over_algorithmic']:.2f}x algorithmic")
    print("wrote", out)


if __name__ == "__main__":
    main()
