"""Multi-GPU: independent-seed environment shards per rank + gradient exchange.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests). Rollout, GAE,
buffer writes and minibatch gradients are rank-local; the only exchange is a SUM all-reduce of the flattened gradient
(actor: 4 610 floats for the 4-64-64-2 net) before Adam, so parameters and Adam state stay replicated.

The reference has no collectives at all; this module is new functionality, defined so that
sync_every == 1 reproduces single-GPU training on the concatenated global minibatch (up to summation order).
"""
import ctypes as C
import math
import numpy as np

from . import _lib as L


def shard_seed(base_seed, rank):
    """Environment seed of a rank: Philox key differs per rank so shards are independent."""
    return (int(base_seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xD1B54A32D192ED03 + 1) & 0xFFFFFFFFFFFFFFFF


def partition_envs(n_envs_global, world_size, rank):
    """Contiguous env ranges [lo, hi) per rank (rank g owns envs g*E/N .. (g+1)*E/N)."""
    per = n_envs_global // world_size
    if per * world_size != n_envs_global:
        raise ValueError("n_envs_global must be divisible by world_size")
    return rank * per, (rank + 1) * per


class GradAllReducer:
    """All-reduces a network's flat gradient in place. The gradient buffer lives in the library (crux_mlp_grads_ptr);
    it is exposed to torch without a copy through __cuda_array_interface__ so RCCL reads/writes it directly."""

    def __init__(self, net, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.net, self.group = torch, dist, net, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.tensor = None

    def _wrap(self):
        if self.tensor is None:
            ptr = self.net.ctx.lib.crux_mlp_grads_ptr(self.net.h)
            n = self.net.n_params

            class _Iface:
                __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}
            self.tensor = self.torch.as_tensor(_Iface(), device="cuda:%d" % self.net.ctx.device)
        return self.tensor

    def __call__(self):
        n_native = self.net.ctx.comm_size()
        if n_native > 1:                    # the library's own RCCL communicator (Context.comm_init): stream-ordered, no host synchronisation
            self.net.ctx.check(self.net.ctx.lib.crux_allreduce_grads(self.net.h))
            return 1.0 / n_native
        if self.world == 1:
            return 1.0
        self.net.ctx.sync()                 # the library's stream must have produced the gradient
        t = self._wrap()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        self.torch.cuda.current_stream().synchronize()
        return 1.0 / self.world            # scale applied inside crux_adam_apply


def train_minibatch_synced(pi, p, P, D, ids0, reducer, cfg_builder):
    """One data-parallel gradient step: local loss+grad -> all-reduce(sum) -> Adam on the mean gradient."""
    cfg = cfg_builder(pi, p, P)
    raw = np.zeros(L.INFO_N, np.float32)
    ids0 = np.ascontiguousarray(ids0, np.int64)
    pi.ctx.check(pi.ctx.lib.crux_loss_grad(pi.h, D.h, C.byref(cfg), ids0.ctypes.data_as(C.c_void_p), ids0.size, raw.ctypes.data_as(C.c_void_p)))
    scale = reducer()
    pi.ctx.check(pi.ctx.lib.crux_adam_apply(pi.h, float(scale)))
    return raw


def sync_due(steps_taken, k):
    """periodic form (crux_peer_set_sync_every(k), bench.py --sync-every k): the replicas average theta, m, v after every k-th minibatch step; k = 1 is the per-step gradient exchange."""
    return k > 1 and steps_taken > 0 and steps_taken % k == 0


class StateAverager:
    """Host-side twin of the in-kernel periodic exchange (train_fs2_kernel.h): replaces every float32 array by its mean over the ranks, in place -- SUM all-reduce, then x float32(1 / N),
    the kernel's arithmetic (at N = 2 the sum is order-free, so the result is bit-identical to the kernel's rank-ordered sum). `arrays` are numpy float32 arrays on the host."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def __call__(self, arrays):
        if self.world == 1:
            return
        inv = np.float32(1.0) / np.float32(self.world)
        for a in arrays:
            t = self.torch.from_numpy(a)                 # aliases the array
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            a *= inv


def train_periodic(local_step, state_arrays, n_steps, k, averager):
    """n_steps local minibatch steps (local_step(i) trains on this rank's shard only), the replicated state averaged after every k-th: the schedule the learner kernel runs under
    crux_peer_set_sync_every(k). Returns the number of exchanges."""
    n_sync = 0
    for i in range(n_steps):
        local_step(i)
        if sync_due(i + 1, k):
            averager(state_arrays()); n_sync += 1
    return n_sync
