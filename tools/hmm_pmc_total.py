#!/usr/bin/env python3
"""HBM bytes per HMM-VB iteration from rocprofv3 --pmc passes over tools/bench_hmm.py.

usage: hmm_pmc_total.py <FETCH_SIZE dir> <WRITE_SIZE dir> --config "K32 D16 T10000000" --json out.json
Sums FETCH_SIZE (KiB, x2 on gfx950: MI355X_MICROARCH.md) and WRITE_SIZE (KiB) over EVERY gmmvb kernel launch of the run
and divides by the number of forward-backward passes (launches of hmm_finish_kernel): the Viterbi pass at the end and the
K-side kernels are left out."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def totals(d):
    per_kernel = defaultdict(float)
    launches = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].replace("void ", "").split("(")[0]
                if not name.startswith("gmmvb::"):
                    continue
                base = name.replace("gmmvb::", "").split("<")[0]
                per_kernel[base] += float(r["Counter_Value"]) * 1024.0
                launches[base] += 1
    return per_kernel, launches


def main():
    a = sys.argv[1:]
    cfg = a[a.index("--config") + 1]
    out = a[a.index("--json") + 1]
    fetch, launches = totals(a[0])
    write, _ = totals(a[1])
    passes = launches.get("hmm_finish_kernel", 0)
    skip = ("hmm_vit", "hmm_viterbi", "kside", "drift", "chol_inv")
    rows = {}
    total = 0.0
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(skip) or passes == 0:
            continue
        b = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) / passes
        rows[k] = dict(bytes_per_iteration=b, launches_per_iteration=launches.get(k, 0) / passes)
        total += b
    json.dump(dict(config=cfg, passes=passes, bytes_per_iteration=total, kernels=rows,
                   note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/bench_hmm.py, KiB -> bytes, "
                        "FETCH x2 (gfx950), summed over all data-pass kernels of an iteration"), open(out, "w"), indent=1, sort_keys=True)
    print(f"{passes} passes, {total/1e9:.2f} GB per iteration")
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["bytes_per_iteration"])[:12]:
        print(f"  {k:32s} {v['bytes_per_iteration']/1e9:8.3f} GB  x{v['launches_per_iteration']:.1f}")


if __name__ == "__main__":
    main()
