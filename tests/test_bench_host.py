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
    j = _launch_check(extra_env={"BENCH_FAKE_MS": "native:0.31,native_push:0.30,rccl:0.29"})
    ex = j["exchange"]
    assert ex["native"]["ms_per_step"] == 0.31 and ex["native_push"]["ms_per_step"] == 0.30 and ex["rccl"]["ms_per_step"] == 0.29 and ex["chosen"] == "rccl"
    assert j["detaches"] == [["detach", "native", False], ["detach", "native_push", False], ["detach", "rccl", False]]
    j = _launch_check(extra_env={"BENCH_FAKE_MS": "native:0.25,native_push:0.27,rccl:0.29"})
    assert j["exchange"]["chosen"] == "native"
    j = _launch_check(extra_env={"BENCH_FAKE_MS": "native:0.25,native_push:0.24,rccl:0.29"})
    assert j["exchange"]["chosen"] == "native_push"                  # (all three exchanges in one run)


def test_a_transport_that_one_rank_cannot_attach_is_skipped_by_the_whole_group():
    """A failure seen by ONE rank (here rank 1's native attach) must send every rank the same way: rank 0, which did attach,
    leaves the broken group, nobody times that transport, the other one supplies the line."""
    j = _launch_check(extra_env={"BENCH_FAKE_FAIL": "native:1"})
    ex = j["exchange"]
    assert "error" in ex["native"] and "1 rank" in ex["native"]["error"] and ex["chosen"] in ("native_push", "rccl") and "ms_per_step" in ex["rccl"]
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


def test_a_transport_that_raises_does_not_cost_the_number_already_measured():
    """the watchdog covers hangs only.  If the SECOND transport raises (a device-side exchange timeout surfacing in
    its timed region, a rendezvous timeout) the first transport's number must still make the line (with nothing measured yet
    run_exchanges re-raises: there is nothing to protect)."""
    j = _launch_check(extra_env={"BENCH_FAKE_RAISE": "rccl:0"})
    ex = j["exchange"]
    assert ex["chosen"] == "native" and "ms_per_step" in ex["native"] and "stand-in" in ex["rccl"]["error"]


def test_live_counter_passes_are_parsed_into_traffic_and_mfma_busy_fraction(tmp_path, monkeypatch):
    """bench.py measures its own HBM traffic and MFMA-busy fraction with rocprofv3 --pmc passes over itself.
    A stand-in `rocprofv3` that writes the csv a real pass would write checks the plumbing: one pass per counter set, medians
    per (kernel, grid), FETCH doubled and KB -> bytes, busy cycles over 4 x busy-CU cycles."""
    import stat
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text("""#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
d = a[a.index("-d") + 1]
os.makedirs(os.path.join(d, "host", "1"), exist_ok=True)
if "--kernel-trace" in a:
    with open(os.path.join(d, "host", "1", "kt_kernel_stats.csv"), "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\\n')
        f.write('"void bp_wgrad_dma<16, 4, 4, 256, false>(MultiArgs)",220,17138000,77900.0,32.8,75641,99561,1482.0\\n')
        f.write('"void bp_gemm<32, 64, 64, 1, 2, true, false, 0, 0>(GemmArgs, EpiArgs)",440,9724000,22100.0,19.2,20840,37561,939.2\\n')
    sys.exit(0)
counters = a[a.index("--pmc") + 1:a.index("--output-format")]
vals = {"FETCH_SIZE": [92000.0, 92700.0, 92800.0], "WRITE_SIZE": [116000.0, 116800.0, 117000.0],
        "SQ_VALU_MFMA_BUSY_CYCLES": [120.0e6, 120.0e6, 120.0e6], "SQ_BUSY_CU_CYCLES": [42.0e6, 43.0e6, 50.0e6]}
with open(os.path.join(d, "host", "1", "p_counter_collection.csv"), "w") as f:
    f.write("Kernel_Name,Grid_Size,Counter_Name,Counter_Value\\n")
    for c in counters:
        for v in vals[c]:
            f.write('"void bp_wgrad_dma<16, 4, 4, 256, false>(MultiArgs)",933888,%s,%r\\n' % (c, v))
            f.write('"void bp_gemm<32, 64, 64, 1, 2, true, false, 0, 0>(GemmArgs, EpiArgs)",65536,%s,%r\\n' % (c, v / 4))
""")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    one = bench.live_pmc(["FETCH_SIZE"])
    assert one["void bp_wgrad_dma<16, 4, 4, 256, false> grid=933888"]["FETCH_SIZE"] == 92700.0
    lc = bench.live_counters()
    assert lc["wgrad_kernel_ns"] == 77900.0 and lc["wgrad_kernel_calls"] == 220      # the headline figure: rocprofv3 --kernel-trace average
    assert lc["traffic_bytes"] == (2 * 92700.0 + 116800.0) * 1024.0
    assert abs(lc["mfma_util"]["wgrad_update_grouped"]["mfma_busy_frac"] - 120.0e6 / (4 * 43.0e6)) < 1e-12
    assert abs(lc["mfma_util"]["hidden_fwd_2048x2048"]["mfma_busy_frac"] - 30.0e6 / (4 * 10.75e6)) < 1e-12
    # no profiler on the box: the line keeps the committed, source-stamped figures
    monkeypatch.setenv("PATH", "/nonexistent")
    monkeypatch.setattr(bench.os.path, "exists", lambda p: False if p.endswith("rocprofv3") else os.path.lexists(p))
    assert bench.live_pmc(["FETCH_SIZE"]) is None
