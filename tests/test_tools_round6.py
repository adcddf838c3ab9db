"""The round-6 measurement tools, on the CPU: the ISA classifier of tools/isa_issue_classes.py against the classes measured in
profiles/r06_issue_rate.md, the issue bound bench.py quotes for the closest-point kernel, and the committed counter traffic the dense
rooflines read."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


def test_instruction_classes_follow_the_measured_table():
    import isa_issue_classes as I
    fast = ["v_fma_f32 v1, v2, v3, v4", "v_add_f32_e32 v1, v2, v3", "v_mul_f32_e32 v1, 2.0, v3", "v_fmac_f32_e32 v1, v2, v3", "v_mov_b32_e32 v1, v2",
            "v_and_b32_e32 v1, v2, v3", "v_add_u32_e32 v1, v2, v3", "v_ashrrev_i32_e32 v1, 31, v2", "v_lshrrev_b32_e32 v1, 4, v2", "v_fma_f32 v1, -v2, |v3|, v4"]
    slow = ["v_max_f32_e32 v1, v2, v3", "v_cmp_lt_f32_e32 vcc, v1, v2", "v_cndmask_b32_e32 v1, v2, v3, vcc", "v_add_f32_e32 v1, s4, v3", "v_mov_b32_e32 v1, s5",
            "v_mov_b32_dpp v1, v2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_lshlrev_b32_e32 v1, 2, v2", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]",
            "v_mad_u32_u24 v1, v2, v3, v4", "v_readfirstlane_b32 s1, v2", "v_cvt_f32_i32_e32 v1, v2", "v_lshl_add_u64 v[0:1], s[2:3], 0, v[4:5]"]
    trans = ["v_rcp_f32_e32 v1, v2", "v_sqrt_f32_e32 v1, v2", "v_readlane_b32 s1, v2, s3"]
    for t in fast:
        assert I.classify(t) == "fast", t
    for t in slow:
        assert I.classify(t) == "slow", t
    for t in trans:
        assert I.classify(t) == "trans", t


def test_closest_point_issue_bound_is_the_busiest_pipe():
    import bench_configs as BC
    s, which, pipes = BC.nearest_issue_bound(83800)
    assert which == max(pipes, key=pipes.get) == "slow vector pipe"
    # by hand: queries x cycles per query of that pipe / (1,024 SIMDs x 2.4 GHz)
    m = BC.NEAREST_MIX
    per_query = BC.NEAREST_VALU_PER_QUERY * (m["slow"] * BC.ISSUE_SLOW + m["trans"] * BC.ISSUE_TRANS + m["vccrun"] * BC.ISSUE_VCCRUN)
    assert s == pytest.approx(83800 * per_query / 1024 / 2.4e9, rel=1e-12)
    assert abs(sum(m.values()) - 1.0) < 1e-9
    # between "every instruction at 2 cycles" and "every instruction at 4"
    assert 83800 * BC.NEAREST_VALU_PER_QUERY * 2 / 1024 / 2.4e9 < s < 83800 * BC.NEAREST_VALU_PER_QUERY * 4 / 1024 / 2.4e9
    d = BC.nearest_dominant(83800, 137e-6)
    assert d["bound"] == "issue: slow vector pipe" and 0.5 < d["frac"] < 0.75


def test_config5_bytes_are_charged_by_iteration_kind():
    import bench_configs as BC
    # 300 iterations: 101 keypoint-only forwards (i <= 100), 199 with forward + full reverse pass (smplify.py:205)
    assert BC.cfg5_bytes_per_frame(300) == 101 * BC.BYTES_SMPLX_FWD + 199 * BC.BYTES_CFG5_ITER
    assert BC.cfg5_bytes_per_frame(300) < 300 * BC.BYTES_CFG5_ITER


def test_committed_counter_traffic_has_the_dense_section():
    with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    assert t["bf_fit_kernel_bytes_per_launch"] < 1e6 < t["bf_mesh_kernel_bytes_per_launch"]
    for cfg, frames, iters in (("cfg3", 1, 200), ("cfg5", 8, 300)):
        d = t["dense"][cfg]
        assert d["frames_per_fit"] == frames and d["iters"] == iters
        assert d["bytes_per_fit"] == pytest.approx(sum(k["launches_per_fit"] * (k["read_bytes_per_launch"] + k["write_bytes_per_launch"]) for k in d["kernels"].values()), rel=1e-9)
        assert any("bf_mesh_multi_kernel" in k for k in d["kernels"]) and any("bf_mesh_bwd_multi_kernel" in k for k in d["kernels"])
    assert any("bf_nearest_kernel" in k for k in t["dense"]["cfg5"]["kernels"])
    import bench_configs as BC
    assert BC.dense_traffic("cfg5", 8, 300) == t["dense"]["cfg5"]["bytes_per_fit"] and BC.dense_traffic("cfg5", 4, 300) is None
