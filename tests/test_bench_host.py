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


def _launch_check(extra_args=(), extra_env=None):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"] + list(extra_args),
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    return _last_json(r.stdout)


def test_multi_gpu_line_times_both_transports_and_takes_the_faster():
    """north_star names RCCL, the library's default transport is its own peer kernels -- `bench.py --gpus N`
    alone must answer which is faster.  The selection flow (run_exchanges) is walked here with stand-in timings: both
    transports attached by the whole group, timed, detached; value = the faster one."""
    j = _launch_check(extra_env={"BENCH_FAKE_MS": "native:0.31,rccl:0.29"})
    ex = j["exchange"]
    assert ex["native"]["ms_per_step"] == 0.31 and ex["rccl"]["ms_per_step"] == 0.29 and ex["chosen"] == "rccl"
    assert j["detaches"] == [["detach", "native", False], ["detach", "rccl", False]]
    j = _launch_check(extra_env={"BENCH_FAKE_MS": "native:0.25,rccl:0.29"})
    assert j["exchange"]["chosen"] == "native"


def test_a_transport_that_one_rank_cannot_attach_is_skipped_by_the_whole_group():
    """A failure seen by ONE rank (here rank 1's native attach) must send every rank the same way: rank 0, which did attach,
    leaves the broken group, nobody times that transport, the other one supplies the line."""
    j = _launch_check(extra_env={"BENCH_FAKE_FAIL": "native:1"})
    ex = j["exchange"]
    assert "error" in ex["native"] and "1 rank" in ex["native"]["error"] and ex["chosen"] == "rccl" and "ms_per_step" in ex["rccl"]
    assert j["detaches"][0] == ["detach", "native", True]          # rank 0's log: it had attached and backed out


def test_a_single_transport_can_still_be_requested():
    j = _launch_check(extra_args=["--exchange", "rccl"])
    assert set(j["exchange"]) == {"rccl", "chosen"} and j["exchange"]["chosen"] == "rccl"


def test_cpu_baseline_carries_the_c1_figure(oracle_mod):
    """CPU baseline for C1 (full) and C2."""
    import bench
    from oracle import oracle as O
    r = bench.cpu_baseline_c1(O, 2, frames=1280, budget_s=1.0)
    assert r["kind"] == "port" and r["unit"] == "frames/s" and r["value"] > 0 and "257->512->257" in r["sample"]


def test_a_transport_that_hangs_does_not_cost_the_number_already_measured():
    """RCCL's bootstrap has never run across devices here: if the second transport hangs inside a foreign call, a watchdog
    thread prints the line from the transport already timed and ends every rank (a signal handler would never get to run)."""
    import time
    t0 = time.time()
    j = _launch_check(extra_env={"BENCH_FAKE_HANG": "rccl:1", "BENCH_WATCHDOG_S": "3"})
    ex = j["exchange"]
    assert ex["chosen"] == "native" and "ms_per_step" in ex["native"] and "watchdog" in ex["rccl"]["error"]
    assert time.time() - t0 < 60
