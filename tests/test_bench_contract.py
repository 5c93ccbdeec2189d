"""bench.py's one-line JSON contract, checked on the line committed under profiles/ (the bench itself needs a GPU)."""
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_committed_bench_line_has_the_contract_fields():
    files = sorted((ROOT / "profiles").glob("r*_bench_n1_v*.json"), key=lambda p: tuple(int(x) for x in re.search(r"r(\d+)_bench_n1_v(\d+)", p.name).groups()))
    line = json.loads(files[-1].read_text().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "GB/s" and line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes"]
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "GB/s" and "sample" in c
    # value = 32 B x Dim / step time
    dim = int(re.search(r"Dim=(\d+)", line["config"]["workload"]).group(1))
    assert abs(line["value"] - 32.0 * dim / (line["ms_per_step"] * 1e-3) / 1e9) < 0.5


def test_bench_defaults_follow_the_measurement_spec():
    src = (ROOT / "bench.py").read_text()
    assert '"--steps", type=int, default=100' in src and '"--warmup", type=int, default=20' in src      # >=20 warm-up, >=100 timed
    assert '"--gpus", type=int, default=1' in src
