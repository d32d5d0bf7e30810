"""CPU tests of bench.py's host-side pieces (no GPU): the algorithmic-work constants and the
bounded CPU baseline leg."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_flops_per_frame_matches_survey_table():
    import bench
    assert bench.flops_per_frame([257, 512, 257]) == 1315840                      # C1
    assert bench.flops_per_frame(bench.LAYERS) == 76648448                        # C2 / C4
    assert bench.flops_per_frame([3084, 2048, 2048, 2048, 257]) == 78753792       # C3
    assert bench.LAYERS == [2827, 2048, 2048, 2048, 257] and bench.BUNCH == 256


def test_cpu_baseline_is_bounded(oracle_mod):
    import bench
    from oracle import bp_numpy as N
    W, b = N.glorot_net(bench.LAYERS, seed=1, beta=0.5)
    r = bench.cpu_baseline(W, b, budget_s=0.5, max_steps=2)
    assert r["kind"] == "port" and r["unit"] == "frames/s" and r["cores"] >= 1 and r["value"] > 0
    assert "steps" in r["sample"]
