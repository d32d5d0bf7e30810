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


def _last_json(txt):
    import json
    return json.loads([ln for ln in txt.strip().splitlines() if ln.startswith("{")][-1])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it must run N ranks (it used to benchmark ONE GPU
    and print n_gpus 1).  --launch-check stops after the ranks have met through the library's rendezvous: no GPU work."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks"] == [0, 1] and j["self_launched"] and len(set(j["pids"])) == 2


def test_bench_refuses_a_rank_count_that_disagrees_with_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_under_torch_distributed_run():
    """The driver's launch line for N > 1 (one rank per GPU from torch.distributed.run): ranks come from the environment."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and not j["self_launched"]


def test_profiled_traffic_is_stamped_with_the_kernel_sources():
    """roofline.traffic comes from a committed PMC pass, not from the run: the line must say which kernel sources the pass
    was taken on and whether they are the ones being run (a stale file must not pass silently)."""
    import bench
    st = bench.kernel_source_stamp()
    assert len(st) == 16 and int(st, 16) >= 0
    t, stamp = bench.profiled_traffic("r02_pmc_hbm_traffic.json", lambda k: "bp_wgrad_dma" in k and "grid=" in k and int(k.split("grid=")[1]) > 500000)
    assert t and 2.5e8 < t < 4e8 and stamp["kernel"].startswith("void bp_wgrad_dma") and stamp["matches_current_kernel_sources"] in (None, False)
    assert bench.profiled_traffic("no_such_file.json", lambda k: True) == (None, None)
