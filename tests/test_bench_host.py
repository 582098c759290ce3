"""Host-side logic of bench.py (no GPU): how an instrumented pass is priced against the roofs, which entry point is
reported as dominant, where counter traffic comes from and when it is refused, the N > 1 self-launch environment."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_layer_table_prices_entry_points_against_their_roofs(monkeypatch):
    monkeypatch.setattr(bench, "_PMC", {"_meta": {"csrc_digest": bench.csrc_digest(), "commit": "abc", "source": "x"},
                                        "entries": {"ecg_conv1d_fwd[256, 128, 256, 125, 15, 7]": 65_000_000}})
    timings = {
        ("ecg_conv1d_fwd", (256, 128, 256, 125, 15, 7)): [0.25, 0.25],                       # ms per call
        ("ecg_conv1d_bwd_weight_bias_ld", (128, 256, 128, 256, 125, 15, 7)): [0.27],         # heaviest: dominant
        ("ecg_conv1d_fwd_bf16_yh", (1, 632, 632, 256, 128, 256, 625, 15, 7)): [0.18],
        ("ecg_conv1d_bwd_data_bf16hh", (640, 632, 256, 128, 256, 625, 15, 7)): [0.16],
        ("ecg_bn_stats_relu_pool_fwd", (1024, 256000, 256, 32, 1000, 0, 0, 0)): [0.014],
    }
    rows, other_ms = bench.layer_table(timings)
    assert abs(other_ms - 0.014) < 1e-12 and len(rows) == 4
    by = {r["entry"].split("[")[0]: r for r in rows}
    f = by["ecg_conv1d_fwd"]
    flops = 2.0 * 256 * 256 * 128 * 15 * 125
    assert f["algorithmic_flops"] == flops and f["calls"] == 2 and f["operands"] == "f32"
    assert abs(f["tflops"] - flops / 0.25e-3 / 1e12) < 0.01 and abs(f["frac"] - f["tflops"] / 157.3) < 1e-3
    assert f["algorithmic_bytes"] == 4.0 * 256 * 125 * (128 + 256) + 4.0 * 256 * 128 * 15
    assert f["traffic_bytes_from_profile"] == 65_000_000 and f["traffic_over_algorithmic"] == round(65e6 / f["algorithmic_bytes"], 3)
    # bf16 activation storage: two bytes per activation element on both sides, priced against the bf16 peak
    h = by["ecg_conv1d_fwd_bf16_yh"]
    assert h["operands"] == "bf16" and h["peak"] == 2500.0
    assert h["algorithmic_bytes"] == 256.0 * (2 * 625 * 128 + 2 * 625 * 256) + 4.0 * 256 * 128 * 15
    assert h["traffic_bytes_from_profile"] is None and h["traffic_source"] == "not collected"
    assert by["ecg_conv1d_bwd_data_bf16hh"]["algorithmic_bytes"] == h["algorithmic_bytes"]
    # the dominant entry point is the one with the largest TOTAL time, multi-launch entry points included
    roof = bench.roofline_of(rows)
    assert roof["kernel"].startswith("ecg_conv1d_fwd[") and roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s"
    timings[("ecg_conv1d_bwd_weight_bias_ld", (128, 256, 128, 256, 125, 15, 7))] = [0.27, 0.27]
    roof = bench.roofline_of(bench.layer_table(timings)[0])
    assert roof["kernel"].startswith("ecg_conv1d_bwd_weight_bias_ld[") and roof["op"] == "wgrad"
    assert bench.roofline_of([]) is None


def test_counter_traffic_is_refused_for_other_kernel_sources(monkeypatch):
    key = "ecg_conv1d_fwd[256, 12, 32, 1000, 15, 7]"
    monkeypatch.setattr(bench, "_PMC", {"_meta": {"csrc_digest": "0" * 16, "commit": "old", "source": "s"}, "entries": {key: 1}})
    tr, why = bench.pmc_traffic(key)
    assert tr is None and why.startswith("stale: collected for csrc 0000000000000000 at old")
    monkeypatch.setattr(bench, "_PMC", {})
    assert bench.pmc_traffic(key) == (None, "not collected")


def test_committed_counter_traffic_matches_the_committed_kernels():
    """profiles/pmc_traffic.json must have been collected for the kernel sources in the tree (bench.py prints null
    otherwise), and must cover every conv entry point of the headline step."""
    meta = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert meta["_meta"]["csrc_digest"] == bench.csrc_digest()
    keys = set(meta["entries"])
    for ci, co, L in ((12, 32, 1000), (32, 64, 500), (64, 128, 250), (128, 256, 125)):
        assert f"ecg_conv1d_fwd[256, {ci}, {co}, {L}, 15, 7]" in keys


def test_percentiles_and_flop_counts():
    p = bench.percentiles([1.0, 2.0, 3.0, 4.0, 5.0])
    assert p["median"] == 3.0 and p["min"] == 1.0 and p["max"] == 5.0 and p["n"] == 5 and p["p10"] <= p["median"] <= p["p90"]
    fwd, step = bench.conv_flops_per_window(1000)
    per = [2 * co * ci * 15 * L for ci, co, L in ((12, 32, 1000), (32, 64, 500), (64, 128, 250), (128, 256, 125))]
    assert fwd == sum(per) and step == 3 * sum(per) - per[0]            # block 0 has no input gradient


def test_flat_adamw_exchange_switches_are_noops_without_a_process_group():
    import torch
    sys.path.insert(0, os.path.join(ROOT, "ptbxl-multimodal_amd"))
    from ecg_hip.optim import FlatAdamW
    lin = torch.nn.Linear(4, 3)
    opt = FlatAdamW(lin.parameters(), lr=1e-3)
    assert opt.set_overlap(True) is False
    assert opt.calibrate_overlap(lambda n: pytest.fail("nothing to calibrate on one rank")) == {"mode": "single", "ms_per_step": {}}
