"""rocprofv3 --pmc counter CSVs of tools/pmc_gemm_step.py -> profiles/<tag>_gemm_pmc_traffic.json (and, with an SQ pass, _gemm_pmc_sq.json).

    python tools/pmc_postprocess_step.py <manifest.json> <fetch csv> <write csv> <tag> [<sq csv>]

Every GEMM kernel row (gemm_256_p8 / gemm_256_grp / gemm_w4p) is matched to the manifest by dispatch order; the LAST of the `reps`
launches of each entry is kept.  Corrections as MI355X_MICROARCH.md 'HBM' prescribes: FETCH_SIZE is in KiB and tallies 128-B
requests at 64 B on gfx950 (x2); WRITE_SIZE in KiB, exact.  bench.py reads the newest *_gemm_pmc_traffic.json."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("gemm_256_p8", "gemm_256_grp", "gemm_w4p")


def rows_of(path, counter):
    out = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNELS)]
    out.sort(key=lambda r: int(r["Dispatch_Id"]))
    return out


def short(kernel):
    for k in KERNELS:
        if k in kernel:
            return k + kernel.split(k, 1)[1].split("(")[0]
    return kernel[:60]


def main():
    manifest = json.load(open(sys.argv[1]))
    fetch_csv, write_csv, tag = sys.argv[2:5]
    f, w = rows_of(fetch_csv, "FETCH_SIZE"), rows_of(write_csv, "WRITE_SIZE")
    n_expected = sum(e["launches"] * e["reps"] for e in manifest)
    assert len(f) == len(w) == n_expected, (len(f), len(w), n_expected)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/pmc_gemm_step.py 3 <manifest>; last launch of each entry",
           "corrections": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md 'HBM'); WRITE_SIZE (KiB) x 1024 exact",
           "note": "TCC_EA fabric requests: Infinity-Cache (MALL) hits are counted, so this is L2-miss traffic, an upper bound on HBM bytes.",
           "kernel": "the launch mix of the 3B step: forward sites through the default algo, backward as grouped dgrad + wgrad launches where the plan groups",
           "sites": {}}
    i = 0
    for e in manifest:
        tags = e["tag"] if isinstance(e["tag"], list) else [e["tag"]]
        algs = e["algorithmic_bytes"] if isinstance(e["algorithmic_bytes"], list) else [e["algorithmic_bytes"]]
        base = i + e["launches"] * (e["reps"] - 1)
        for j, (t, a) in enumerate(zip(tags, algs)):
            fr, wr = f[base + j], w[base + j]
            fk, wk = float(fr["Counter_Value"]), float(wr["Counter_Value"])
            out["sites"][t] = {"site": e["site"], "kernel": short(fr["Kernel_Name"]), "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                               "fabric_read_bytes": fk * 1024 * 2, "fabric_write_bytes": wk * 1024, "algorithmic_bytes": a}
        i += e["launches"] * e["reps"]
    prof = os.path.join(ROOT, "profiles")
    json.dump(out, open(os.path.join(prof, f"{tag}_gemm_pmc_traffic.json"), "w"), indent=1)
    for name, rows in (("fetch_size", f), ("write_size", w)):
        with open(os.path.join(prof, f"{tag}_gemm_pmc_{name}.csv"), "w", newline="") as fh:
            wr_ = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            wr_.writeheader()
            wr_.writerows(rows)
    for t, v in out["sites"].items():
        print(f"{t:40s} {v['kernel'][:28]:28s} {v['site']:44s} fabric {(v['fabric_read_bytes'] + v['fabric_write_bytes']) / 1e6:8.1f} MB = "
              f"{(v['fabric_read_bytes'] + v['fabric_write_bytes']) / v['algorithmic_bytes']:.2f} x algorithmic")
    if len(sys.argv) > 5:
        sq = [r for r in csv.DictReader(open(sys.argv[5])) if any(k in r["Kernel_Name"] for k in KERNELS)]
        by_disp = {}
        for r in sq:
            by_disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
        disp = sorted(by_disp)
        assert len(disp) == n_expected, (len(disp), n_expected)
        res, i = {}, 0
        for e in manifest:
            tags = e["tag"] if isinstance(e["tag"], list) else [e["tag"]]
            base = i + e["launches"] * (e["reps"] - 1)
            for j, t in enumerate(tags):
                c = by_disp[disp[base + j]]
                wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
                res[t] = {"site": e["site"], "kernel": short(c["kernel"]), **{k: v for k, v in c.items() if k != "kernel"},
                          "wave_cycles_waiting_frac": c.get("SQ_WAIT_ANY", 0.0) / wc, "wave_cycles_issue_stalled_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                          "wave_cycles_issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
            i += e["launches"] * e["reps"]
        json.dump({"source": "rocprofv3 --pmc SQ_* (own pass) -- python3 tools/pmc_gemm_step.py 3", "sites": res},
                  open(os.path.join(prof, f"{tag}_gemm_pmc_sq.json"), "w"), indent=1)
        with open(os.path.join(prof, f"{tag}_gemm_pmc_sq.csv"), "w", newline="") as fh:
            wr_ = csv.DictWriter(fh, fieldnames=list(sq[0].keys()))
            wr_.writeheader()
            wr_.writerows(sq)


if __name__ == "__main__":
    main()
