"""CPU-side checks of the measurement plumbing: the synthetic workload generator, the profile-summary parser and the
committed bench lines (the numbers themselves are measured on the MI355X; here only their consistency)."""
import glob
import json
import os

import numpy as np

import bench
from gradslam_amd.datasets import synthetic

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_camera_path_turns_back_and_early_frames_are_unchanged():
    # the goldens were recorded on the first frames of the one-way path: identity there
    for s in range(0, synthetic.PATH_TURN + 1, 7):
        assert synthetic.path_parameter(s) == s
    assert synthetic.path_parameter(synthetic.PATH_TURN + 10) == synthetic.PATH_TURN - 10
    assert synthetic.path_parameter(2 * synthetic.PATH_TURN) == 0
    assert np.array_equal(synthetic.gt_pose(2 * synthetic.PATH_TURN + 3), synthetic.gt_pose(3))
    yaw = max(abs(np.arctan2(synthetic.gt_pose(s)[0, 2], synthetic.gt_pose(s)[0, 0])) for s in range(0, 1000, 13))
    assert yaw <= 0.3 + 1e-6


def test_chunked_generation_is_a_function_of_the_frame_index():
    """bench.make_sequences builds long sequences from chunks generated in parallel: geometry (depth before the
    holes, poses) of a frame must not depend on the chunking."""
    a = synthetic.make_sequence(4, 24, 32, seed=2, first=8)
    b = synthetic.make_sequence(2, 24, 32, seed=2, first=10)
    assert np.array_equal(a["poses"][2:], b["poses"])
    both = (a["depths"][2:] > 0) & (b["depths"] > 0)    # the hole streams differ per chunk
    assert both.mean() > 0.8 and np.array_equal(a["depths"][2:][both], b["depths"][both])


def test_sequences_generated_in_process_under_a_profiler(monkeypatch):
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    seqs = bench.make_sequences([0, 1], 2, 24, 32)
    assert len(seqs) == 2 and seqs[0]["depths"].shape == (2, 24, 32, 1)
    ref = synthetic.make_sequence(2, 24, 32, seed=1)
    assert np.array_equal(seqs[1]["depths"], ref["depths"])


def test_pmc_summary_parser_and_committed_bench_lines():
    traffic, src = bench.pmc_traffic("gs_icp_half_batch_kernel")
    assert src is not None and src["file"].startswith("profiles/") and 1e6 < traffic < 1e9
    assert src["command"] and "bench.py" in src["command"]          # the run the counters came from, on record
    lines = sorted(glob.glob(os.path.join(REPO, "profiles", "r02_*_bench_line.json")))
    assert lines
    for path in lines:
        d = json.load(open(path))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config"):
            assert k in d, (path, k)
        assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
        # value = sequences * steps / time
        seqs = d["config"]["sequences_total"]
        assert abs(d["value"] - seqs * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
        r = d.get("roofline")
        if r:
            assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
            if r.get("traffic") is not None:
                assert r["alg_le_traffic"] == (r["alg_bytes_per_launch"] <= r["traffic"])
