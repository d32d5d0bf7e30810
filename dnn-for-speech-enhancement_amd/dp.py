"""Data-parallel driver: one process per GPU, minibatch sharded over ranks, one gradient
exchange per minibatch (semantics donor = the reference's commented-out
train_bunch_multi, BP_GPU.cu:775-908: per-GPU fwd/bwd, gradient SUM, one update with
n = global bunch, identical state on every rank afterwards).

The step engine is pluggable only so that the exchange logic can be exercised by CPU tests
with a `gloo` process group; the product engine is `HipEngine` (libbp_hip.so) and nothing here
falls back to a CPU computation.
"""
import numpy as np


class HipEngine(object):
    """Per-rank compute engine on the HIP library: gradients of the local shard go to a torch
    tensor that RCCL reduces in place; all device work runs on torch's current stream so the
    collective is ordered against it."""

    def __init__(self, pkg, layersizes, local_bunch, world, rank, lrate, momentum, weightcost, weights, bias,
                 device=0, max_chunk_frames=0, **kw):
        import torch
        self.torch = torch
        self.world, self.rank, self.B = world, rank, local_bunch
        self.obj = pkg.BP_GPU(world, len(layersizes), layersizes, local_bunch, lrate, momentum, weightcost, weights,
                              bias, device=device, global_bunchsize=local_bunch * world,
                              rank_frame_offset=rank * local_bunch, max_chunk_frames=max_chunk_frames, **kw)
        n = self.obj.grad_floats()
        self.grad = torch.zeros(n, dtype=torch.float32, device="cuda:%d" % device)
        self.obj.use_grad_buffer(self.grad.data_ptr(), n)
        self.obj.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self.segments = [self.obj.grad_layout(l) for l in range(1, len(layersizes))]

        self.nlayers = len(layersizes)

    def grads(self, first_frame):
        self.obj.grads_resident(first_frame)
        return self.grad

    def update(self):
        self.obj.apply_update()

    # layer-by-layer interface (exchange of layer l overlaps the backward of the layers below)
    def forward(self, first_frame):
        self.obj.dp_forward(first_frame)

    def backward_layer(self, l):
        self.obj.dp_backward_layer(l)
        off, cnt = self.segments[l - 1]
        return self.grad[off:off + cnt]

    def update_layer(self, l):
        self.obj.apply_update_layer(l)

    def advance(self):
        self.obj.advance_step()

    # finer split for the cross-step pipeline
    def forward_layer(self, first_frame, l):
        self.obj.dp_forward_layer(first_frame, l)

    def dgrads(self):
        self.obj.dp_dgrads()

    def wgrad_layer(self, l):
        self.obj.dp_wgrad_layer(l)
        off, cnt = self.segments[l - 1]
        return self.grad[off:off + cnt]


def dp_step(engine, dist, first_frame):
    """One data-parallel minibatch: local gradients -> all-reduce(SUM) -> identical update."""
    g = engine.grads(first_frame)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
    engine.update()


def dp_step_overlapped(engine, dist, first_frame):
    """One data-parallel minibatch with the exchange pipelined against the backward pass: as soon
    as layer l's [W|b] gradient segment is complete its all-reduce is launched (async; RCCL runs
    it on its own stream, ordered after the producing kernels) while dgrad/wgrad of the layers
    below continue; each layer is updated when its sum has landed.  Same arithmetic as dp_step."""
    engine.forward(first_frame)
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    works = []
    for l in range(engine.nlayers - 1, 0, -1):
        seg = engine.backward_layer(l)
        works.append((l, dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True) if multi else None))
    for l, w in works:
        if w is not None:
            w.wait()
        engine.update_layer(l)
    engine.advance()


class DPPipeline(object):
    """Data-parallel training with the exchange pipelined ACROSS steps.  Per bunch: forward
    (layer by layer), every dgrad, then the weight gradients largest-first (layer 1 holds 40 % of
    the bytes), each followed at once by its async all-reduce.  The update of layer l is applied
    right before the NEXT bunch's forward of layer l, so the all-reduces of layers 2.. overlap the
    next forward and only layer 1's is on the critical path.  Arithmetic is unchanged: every
    forward/dgrad of a bunch sees weights that include all updates of the previous bunch, and
    dgrads of a bunch never see its own updates (same as BP_GPU.cu:588-671).  Call flush() after
    the last bunch."""

    def __init__(self, engine, dist):
        self.engine, self.dist = engine, dist
        self.multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
        self.pending = None

    def step(self, first_frame):
        eng, L = self.engine, self.engine.nlayers
        pend = self.pending
        if pend is not None:
            eng.advance()
        for l in range(1, L):
            if pend is not None:
                if pend[l] is not None:
                    pend[l].wait()
                eng.update_layer(l)
            eng.forward_layer(first_frame, l)
        eng.dgrads()
        self.pending = {}
        for l in range(1, L):
            seg = eng.wgrad_layer(l)
            self.pending[l] = (self.dist.all_reduce(seg, op=self.dist.ReduceOp.SUM, async_op=True)
                               if self.multi else None)

    def flush(self):
        if self.pending is None:
            return
        for l in range(1, self.engine.nlayers):
            if self.pending[l] is not None:
                self.pending[l].wait()
            self.engine.update_layer(l)
        self.engine.advance()
        self.pending = None


def shard_rows(n_frames, global_bunch, world, rank):
    """Row indices of `rank`'s shard of every full global bunch of a chunk (partial last bunch
    dropped as in BP_GPU.cu:315-318): bunch i -> rows [i*Bg + r*Bg/G, i*Bg + (r+1)*Bg/G)."""
    assert global_bunch % world == 0
    lb = global_bunch // world
    nb = n_frames // global_bunch
    idx = (np.arange(nb)[:, None] * global_bunch + rank * lb + np.arange(lb)[None, :]).reshape(-1)
    return idx
