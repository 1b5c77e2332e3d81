#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters for the gmmvb kernels (largest grid of each kernel only).

usage: summarize_pmc.py <dir> [<dir> ...]     (each dir = one --pmc pass, *_counter_collection.csv inside)
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section); on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced streaming reads, so both the raw and the x2 figure are printed.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    json_out = None
    if "--json" in sys.argv:
        i = sys.argv.index("--json")
        json_out = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    config = None
    if "--config" in sys.argv:          # stamp of the workload the passes were made on, e.g. "K64 D128 N10000000 f32"
        i = sys.argv.index("--config")
        config = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    window = None
    if "--window" in sys.argv:          # e.g. "w5s20": the bench command's warm-up / timed steps the passes were made with
        i = sys.argv.index("--window")
        window = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    rows = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> values (largest grid only)
    seq = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> (dispatch id, value), EVERY launch (any grid)
    grid = {}
    dur = defaultdict(list)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    name = r["Kernel_Name"].replace("void ", "").split("(")[0]
                    if not name.startswith("gmmvb::"):
                        continue
                    g = int(r["Grid_Size"])
                    if g > grid.get(name, 0):
                        grid[name] = g
                        rows[name].clear()
                        dur[name].clear()
                    seq[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                    if g == grid[name]:
                        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for name in sorted(rows, key=lambda n: -sum(dur[n])):
        print(f"## {name}  (grid {grid[name]} threads, profiled launch {sum(dur[name])/len(dur[name]):.2f} ms avg)")
        for c, v in sorted(rows[name].items()):
            avg = sum(v) / len(v)
            extra = ""
            if c == "FETCH_SIZE":
                extra = f"  = {avg*1024/1e9:.3f} GB raw, {2*avg*1024/1e9:.3f} GB with the gfx950 x2 correction"
            if c == "WRITE_SIZE":
                extra = f"  = {avg*1024/1e9:.3f} GB"
            print(f"  {c:32s} {avg:18.1f}   (n={len(v)}){extra}")
        c = {k: sum(v) / len(v) for k, v in rows[name].items()}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
            util = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0)
            print(f"  => MFMA pipe utilisation = busy cycles per SIMD / kernel cycles = {100*util:.1f} %")
            if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
                print(f"  => executed f64 MFMA flops = MOPS x 512 = {c['SQ_INSTS_VALU_MFMA_MOPS_F64']*512/1e12:.3f} TFLOP per launch")
            if "SQ_INSTS_VALU" in c:
                n_mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 64.0
                print(f"  => non-MFMA vector instructions per MFMA = {(c['SQ_INSTS_VALU'] - n_mfma) / n_mfma:.2f}")
        print()
    if json_out:
        import json
        out = {}
        # template variants of one kernel (e.g. the gather with and without the early way out) share an entry: the
        # average is over all their launches
        merged = defaultdict(lambda: defaultdict(list))
        names = defaultdict(list)
        for name in rows:
            base = name.replace("gmmvb::", "").split("<")[0]
            names[base].append(name)
            for k, v in rows[name].items():
                merged[base][k].extend(v)
        for base in merged:
            c = {k: sum(v) / len(v) for k, v in merged[base].items()}
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                # per launch, in dispatch order (the template variants of a kernel interleaved as they ran): lets bench.py
                # take exactly the launches of its timed steps when it runs the command the passes were made with
                per = {}
                for cn in ("FETCH_SIZE", "WRITE_SIZE"):
                    ent = sorted(v for n in names[base] for v in seq[n][cn])
                    per[cn] = [round(v * 1024) for _d, v in ent]
                out[base] = dict(
                    window=window, fetch_bytes_raw_launches=per["FETCH_SIZE"], write_bytes_launches=per["WRITE_SIZE"],
                    kernel=" + ".join(sorted(names[base])), grid_threads=max(grid[n] for n in names[base]),
                    launches=len(merged[base]["FETCH_SIZE"]), fetch_bytes_raw=c["FETCH_SIZE"] * 1024,
                    fetch_bytes=2 * c["FETCH_SIZE"] * 1024, write_bytes=c["WRITE_SIZE"] * 1024, config=config,
                    note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB -> bytes, FETCH x2 (gfx950)")
        with open(json_out, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
