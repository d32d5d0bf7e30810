#!/usr/bin/env python3
"""bench.py -- training frames/sec of the frame-wise DNN step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload at every N (weak scaling, per-GPU work fixed): BASELINE.json configs[1] = C2:
2827->2048->2048->2048->257 (257x11 stacked input), ReLU + dropout 0.1/0.2, fp32, 256 frames
per GPU per step, lrate 1, momentum 0.5, synthetic N(0,1) frames resident in HBM (generated on
device), Glorot*0.5 weights.  A step = forward + backward + momentum update of one bunch
(train_bunch_single, BP_GPU.cu:484-673); for N>1 the global bunch is N*256 frames with one
RCCL all-reduce(SUM) of the gradients per step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: needed by RCCL across processes on this driver

import numpy as np  # noqa: E402

LAYERS = [257 * 11, 2048, 2048, 2048, 257]
BUNCH = 256
CHUNK = 102400            # frames resident per chunk (finetune_..._NAT.pl:39 traincache)
PEAK_MFMA_F32_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def flops_per_frame(ls):
    P = sum(ls[i - 1] * ls[i] for i in range(1, len(ls)))
    return 6 * P - 2 * ls[0] * ls[1]


def cpu_baseline(W, b, budget_s=12.0, max_steps=64):
    """The oracle (C restatement, OpenMP) timed on this box's host cores on a bounded sample of
    the same workload: whole C2 training steps (dropout on) until ~budget_s of CPU time."""
    from oracle import oracle as O
    O.build()
    o = O.Oracle(LAYERS, BUNCH, 1.0, 0.5, 0.0, W, b, dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=1)
    rng = np.random.default_rng(20260927)
    x = rng.standard_normal((BUNCH, LAYERS[0]), dtype=np.float32)
    t = rng.standard_normal((BUNCH, LAYERS[-1]), dtype=np.float32)
    o.train_bunch(x, t)                      # warm-up (page in, thread pool)
    n, t0 = 0, time.perf_counter()
    while n < max_steps and (time.perf_counter() - t0) < budget_s:
        o.train_bunch(x, t)
        n += 1
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": n * BUNCH / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d full C2 training steps (256 frames each, fp32, dropout on) of oracle/bp_oracle.c "
                      "(OpenMP, %d threads), %.1f s" % (n, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=CHUNK)
    args = ap.parse_args()

    import torch
    import dnnse_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_dp = os.environ.get("BENCH_FORCE_DP") == "1"      # exercise the data-parallel path at world size 1
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N"
    dev = local_rank

    W, b = dnnse_amd.glorot_net(LAYERS, seed=1, beta=0.5)    # Gen_rand_net flag=1, beta=0.5 recipe
    chunk = max(BUNCH, (args.chunk // BUNCH) * BUNCH)
    kw = dict(dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=20260927, device=dev, max_chunk_frames=chunk)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if world == 1 and not force_dp:
        g = dnnse_amd.BP_GPU(1, len(LAYERS), LAYERS, BUNCH, 1.0, 0.5, 0.0, W, b, **kw)
        g.fill_chunk_synthetic(chunk, 20260927)
        g.sync()
        nb_chunk = chunk // BUNCH

        def run(nsteps, pos):
            done = 0
            while done < nsteps:
                k = min(nsteps - done, nb_chunk - pos)
                g.train_resident(pos * BUNCH, k * BUNCH)
                done += k
                pos = (pos + k) % nb_chunk
            return pos

        pos = run(args.warmup, 0)
        g.sync(); barrier()
        t0 = time.perf_counter()
        pos = run(args.steps, pos)
        g.sync(); barrier()
        dt = time.perf_counter() - t0
        obj = g
    else:
        from importlib import import_module
        dp = import_module("dnn_for_speech_enhancement_amd.dp")
        eng = dp.HipEngine(dnnse_amd, LAYERS, BUNCH, world, rank, 1.0, 0.5, 0.0, W, b, **kw)
        eng.obj.fill_chunk_synthetic(chunk, 20260927 + rank)       # each rank holds its own shard of every bunch
        nb_chunk = chunk // BUNCH
        pos = 0
        mode = os.environ.get("BENCH_DP_MODE", "pipeline")     # pipeline | overlapped | serial
        if mode == "pipeline":
            pipe = dp.DPPipeline(eng, dist)
            step_fn = lambda e, d, f: pipe.step(f)
            flush = pipe.flush
        else:
            step_fn = dp.dp_step if mode == "serial" else dp.dp_step_overlapped
            flush = lambda: None
        for _ in range(args.warmup):
            step_fn(eng, dist, pos * BUNCH); pos = (pos + 1) % nb_chunk
        flush()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn(eng, dist, pos * BUNCH); pos = (pos + 1) % nb_chunk
        flush()
        barrier()
        dt = time.perf_counter() - t0
        obj = eng.obj

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda:%d" % dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    frames = args.steps * BUNCH * world
    value = frames / dt
    res = {
        "metric": "training frames/sec (257x11 input, 3x2048 DNN)", "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: 2827->2048->2048->2048->257 ReLU+dropout(0.1/0.2), fp32, %d frames/GPU/step "
                               "(global bunch %d), lrate 1, momentum 0.5, %d-frame chunk resident in HBM"
                               % (BUNCH, BUNCH * world, chunk),
                   "parallelism": "dp%d" % world, "frames_per_gpu_per_step": BUNCH, "global_bunch": BUNCH * world},
    }
    if rank == 0:
        # ---- roofline of the dominant kernel: the 2048x2048 hidden-layer forward GEMM
        # (M=256 frames, N=K=2048): algorithmic FLOPs = 2*M*N*K per launch, duration = HIP events
        # around `iters` back-to-back launches on the kernel's own stream (bp_time_kernel).
        it = 200
        ms = {name: obj.time_kernel(k, it) for name, k in
              (("fwd_hidden", 0), ("dgrad_hidden", 1), ("wgrad_update_hidden", 2), ("fwd_l1", 3), ("fwd_out", 4),
               ("wgrad_update_l1", 5))}
        fl = 2.0 * BUNCH * 2048 * 2048
        ach = fl / (ms["fwd_hidden"] * 1e-3) / 1e12
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (separate
        # FETCH_SIZE / WRITE_SIZE runs of this same command, KB units, FETCH doubled per the gfx950
        # note in MI355X_MICROARCH.md): profiles/r01_pmc_hbm_traffic.json.  null when absent.
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
            for k, v in pm.get("kernels", pm).items():
                if "bp_gemm<32, 64, 64, 1, 2, true, false, 0, 1, 0, 0>" in k:      # hidden-layer forward (TAG 0)
                    traffic = (v["fetch_MB_corrected_x2"] + v["write_MB"]) * 1e6
        except Exception:
            traffic = None
        res["roofline"] = {"bound": "mfma", "kernel": "bp_gemm<32,64,64,...,EPI_FWD_HIDDEN> (2048x2048 hidden fwd)",
                           "achieved": ach, "peak": PEAK_MFMA_F32_TF, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F32_TF,
                           "traffic": traffic, "traffic_unit": "bytes/launch (PMC, profiles/r01_pmc_hbm_traffic.json)",
                           "algorithmic_bytes": 4.0 * (2048 * 2048 + 2 * BUNCH * 2048), "kernel_ms": ms,
                           # skinny layers: achieved GB/s = 4*(prev*cur + B*prev + B*cur) / t
                           "skinny_layers_GBs": {
                               "fwd_l1_2827x2048": 4.0 * (LAYERS[0] * LAYERS[1] + BUNCH * (LAYERS[0] + LAYERS[1])) / (ms["fwd_l1"] * 1e-3) / 1e9,
                               "fwd_out_2048x257": 4.0 * (LAYERS[-2] * LAYERS[-1] + BUNCH * (LAYERS[-2] + LAYERS[-1])) / (ms["fwd_out"] * 1e-3) / 1e9},
                           "step_frac_of_mfma_peak": flops_per_frame(LAYERS) * value / world / 1e12 / PEAK_MFMA_F32_TF}
        if world == 1 and not force_dp and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(W, b)
        try:                                   # anything native code left in C stdio (e.g. the RCCL banner) goes out first,
            import ctypes                      # so that the JSON line is the last thing on stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
