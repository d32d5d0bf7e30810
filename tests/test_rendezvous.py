"""CPU tests (no GPU) of the data-parallel ranks' rendezvous -- bp_rdv_* in include/bp_c_api.h, the host half of
bp_dp_attach and of the N-rank launch paths (bench.py --gpus N, bptrain gpu_used=N): world_size 2 and 3, separate
processes, barrier + all-gather, and the lifetime rules: nothing is left in /dev/shm once everyone has joined, a stale
block of a crashed job with the same key is replaced, a missing rank makes the others fail instead of hang."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import sys, time
sys.path.insert(0, %r)
import dnnse_amd
key, world, rank, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
r = dnnse_amd.Rendezvous(key, world, rank, timeout_s=float(sys.argv[5]))
if mode == "crash_after_join" and rank == 1:
    import os; os._exit(7)                      # dies without closing: peers must fail fast or time out, never hang
vals = r.allgather_f64(10.0 * rank + 1.0)
for _ in range(3):
    r.barrier()
mx = max(r.allgather_f64(0.5 + rank))
print("OK", rank, vals, mx, flush=True)
r.close()
""" % ROOT


def _spawn(key, world, rank, mode="ok", timeout_s=20.0):
    return subprocess.Popen([sys.executable, "-c", WORKER, key, str(world), str(rank), mode, str(timeout_s)],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_meet_barrier_and_allgather(pkg, world):
    key = "t-rdv-%d-%d" % (os.getpid(), world)
    ps = [_spawn(key, world, r) for r in range(world)]
    outs = [p.communicate(timeout=60)[0] for p in ps]
    for r, (p, o) in enumerate(zip(ps, outs)):
        assert p.returncode == 0, o
        assert "OK %d %s %s" % (r, [10.0 * q + 1.0 for q in range(world)], world - 0.5) in o, o
    assert not os.path.exists("/dev/shm/bpdp-" + key)          # the name goes as soon as everyone has joined


def test_stale_block_of_a_dead_job_is_replaced(pkg):
    """A block left by a crashed job (creator dead, abort flag set or not) must not poison the next job with its key."""
    key = "t-rdv-stale-%d" % os.getpid()
    path = "/dev/shm/bpdp-" + key
    with open(path, "wb") as f:                                  # garbage of the right size, from a "previous job"
        f.write(b"\x01" * 16384)
    # rank 1 first (it sees the stale block and must keep waiting for the real one), rank 0 a moment later
    p1 = _spawn(key, 2, 1)
    time.sleep(0.5)
    p0 = _spawn(key, 2, 0)
    o0, o1 = p0.communicate(timeout=60)[0], p1.communicate(timeout=60)[0]
    assert p0.returncode == 0 and p1.returncode == 0, (o0, o1)
    assert "OK 0" in o0 and "OK 1" in o1
    assert not os.path.exists(path)


def test_missing_rank_times_out_with_an_error(pkg):
    key = "t-rdv-miss-%d" % os.getpid()
    t0 = time.time()
    p0 = _spawn(key, 2, 0, timeout_s=1.5)                        # rank 1 never comes
    o0 = p0.communicate(timeout=60)[0]
    assert p0.returncode != 0 and "timed out" in o0, o0
    assert time.time() - t0 < 30
    assert not os.path.exists("/dev/shm/bpdp-" + key)           # rank 0 removes the name on the failure path too


def test_dead_peer_makes_the_others_fail_not_hang(pkg):
    key = "t-rdv-dead-%d" % os.getpid()
    ps = [_spawn(key, 2, r, mode="crash_after_join", timeout_s=2.0) for r in range(2)]
    o0 = ps[0].communicate(timeout=60)[0]
    ps[1].communicate(timeout=60)
    assert ps[1].returncode == 7
    assert ps[0].returncode != 0 and ("timed out" in o0 or "peer rank failed" in o0), o0


def test_argument_errors(pkg):
    with pytest.raises(pkg.BPError):
        pkg.Rendezvous("a/b", 2, 0, timeout_s=1.0)
    with pytest.raises(pkg.BPError):
        pkg.Rendezvous("x", 9, 0, timeout_s=1.0)
    with pytest.raises(pkg.BPError):
        pkg.Rendezvous("x", 2, 2, timeout_s=1.0)
    r = pkg.Rendezvous("t-rdv-solo-%d" % os.getpid(), 1, 0, timeout_s=1.0)   # a one-rank group is legal
    assert r.allgather_f64(3.0) == [3.0]
    r.close()


def test_a_second_job_cannot_take_over_the_key_of_a_job_that_is_still_joining(pkg):
    """rank 0 used to unlink whatever carried the name; a second job started with the same key would then
    capture the first job's late ranks.  Now the second creator is refused while the first block's creator is alive."""
    key = "t-rdv-live-%d" % os.getpid()
    p0 = _spawn(key, 2, 0, timeout_s=15.0)                       # job A, waiting for its rank 1
    deadline = time.time() + 10
    while not os.path.exists("/dev/shm/bpdp-" + key) and time.time() < deadline:
        time.sleep(0.05)
    time.sleep(0.2)
    q0 = _spawn(key, 2, 0, timeout_s=3.0)                        # job B, same key
    oq = q0.communicate(timeout=60)[0]
    assert q0.returncode != 0 and "live job" in oq, oq
    p1 = _spawn(key, 2, 1, timeout_s=15.0)                       # job A's late rank still finds job A
    o0, o1 = p0.communicate(timeout=60)[0], p1.communicate(timeout=60)[0]
    assert p0.returncode == 0 and p1.returncode == 0, (o0, o1)
    assert "OK 0" in o0 and "OK 1" in o1


def test_duplicate_rank_is_an_error(pkg):
    key = "t-rdv-dup-%d" % os.getpid()
    p0 = _spawn(key, 3, 0, timeout_s=4.0)
    p1 = _spawn(key, 3, 1, timeout_s=4.0)
    time.sleep(1.0)
    d1 = _spawn(key, 3, 1, timeout_s=4.0)                        # a second process claiming rank 1
    od = d1.communicate(timeout=60)[0]
    assert d1.returncode != 0 and "already taken" in od, od
    p0.communicate(timeout=60); p1.communicate(timeout=60)       # (rank 2 never comes: they time out)
    assert p0.returncode != 0 and p1.returncode != 0
