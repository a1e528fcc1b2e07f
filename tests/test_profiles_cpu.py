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
    v = bench.pmc_valu_issue_ntt()
    assert v is not None and 0.5 < v < 0.9  # a two-generation launch: fill and drain cost a quarter of it


import pytest


@pytest.mark.parametrize("name", ["r02_bench_create_proof_k18_line_unprofiled.json", "r03_bench_create_proof_k18_line_unprofiled.json"])
def test_committed_bench_line_has_the_contract_fields(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
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
    if name.startswith("r03"):  # the legs SURVEY 8(d) asks for beside the headline
        assert {"2^16", "2^18", "2^20", "2^22"} <= set(d["msm_standalone"]) and d["msm_standalone"]["2^18"]["single_mscalar_per_s"] > 100
        assert d["ntt_ext"]["coeff_to_extended"]["melem_per_s"] > 1000 and d["host_advice"]["proof_equals_device_resident"] is True
        assert d["batched_k22"]["k"] == 22 and d["batched_k22"]["first_proof_equals_single"] is True and "dense_witness" in d
        assert d["batched"]["lanes_3"]["reported"].startswith("median") and d["dense_equivalent_mscalar_per_s"] < d["value"]
