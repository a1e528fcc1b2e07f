"""CPU: the committed rocprofv3 summaries that bench.py quotes (`roofline.traffic`, `roofline.valu_issue_utilisation_pmc`)
parse, and the committed bench line carries the fields the driver's contract names."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_summaries_parse():
    t = bench.pmc_traffic("msm_accumulate_kernel")
    assert t is not None and t > 1e8  # hundreds of MB per launch
    assert bench.pmc_traffic("ntt_pass_kernel<6u, 4u>") > 1e7
    u = bench.pmc_valu_issue()
    assert u is not None and 0.5 < u <= 1.0


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r02_bench_create_proof_k18_line_unprofiled.json")) as f:
        d = json.load(f)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["proof_bytes_identical"] is True
    assert d["cpu_baseline"]["k"] == 18 and d["cpu_baseline"]["sample_small"]["k"] == 16  # the metric's own configuration
    assert d["roofline"]["bound"] == "valu" and "not this run" in d["roofline"]["traffic_source"]
    assert "per_word_callback" in d["generic_rng"] and "unsharded" in d["config3_k20"] and "lanes_3" in d["batched"]
