"""The ONE JSON line bench.py prints is what the driver records: its keys (the contract's, `roofline` with both fractions,
`cpu_baseline`, the sub-lines) and its size (< 6 KB: round 4's 22-KB line did not parse in the driver's record) are pinned here
on a full record of a real run (profiles/r6_bench_detail_sample.json, made on the GPU by `bench.py --detail`), without a GPU."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record():
    with open(os.path.join(ROOT, "profiles", "r6_bench_detail_sample.json")) as f:
        return json.load(f)


def test_compact_line_has_the_contracts_keys_and_fits():
    out = _record()
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < 6000, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"].startswith("GMM-VB E+M samples/sec at K=64,D=128,N=1e7") and line["unit"] == "samples/s"
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic" and line["higher_is_better"] is True
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_executed", "frac_on_F", "step_frac_on_F", "pruned"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] <= 1.0 and r["frac"] == r["frac_executed"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["pruned"] is True and r["frac_on_F"] > 1.0            # the F-basis fraction of a pruned pass is not a utilisation
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "threads_swept" in c
    for sub in ("hmm_c5", "c2", "c4_shard", "c4_strong"):
        assert sub in line, sub
    for sub in ("c2", "c4_shard", "c4_strong"):
        for key in ("ms_per_step", "kernel", "frac_executed", "frac_on_F", "workload"):
            assert key in line[sub], (sub, key)
    assert line["parity"]["passed"] and line["parity_sparse_path"]["passed"]
    assert line["parity"]["max_rel_err"] < 1e-5 and line["parity"]["tolerance"] == 1e-5


def test_value_is_rows_over_time():
    out = _record()
    assert abs(out["value"] - out["config"]["rows_total"] / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
