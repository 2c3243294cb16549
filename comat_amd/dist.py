"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm;
"gloo" for the CPU tests).  The ONLY exchange of the CoMat step is an all-reduce(mean) of the flat fp32 trainable
gradient buffer, once per optimizer step for G and once for D (the reference's DDP all-reduce,
training_script.py:659,690).  The buffers are single contiguous tensors (102 MB for SD1.5 LoRA r=128), so each
all-reduce is one large collective — sized for xGMI's per-link-bound ring, not bucketed into small messages — and
it is launched asynchronously so that the D-step forward/backward overlaps the G-gradient reduction."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from torchrun-style env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Returns (rank, world, dev)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, (torch.device(f"cuda:{local}") if use_gpu else torch.device("cpu"))


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


class GradReducer:
    """all-reduce of flat fp32 gradient buffers, asynchronous.  The collective is a SUM; `finish()` returns the factor
    1 / world that makes it the mean of the ranks' gradients (DDP semantics) and the optimizer kernel applies it to the
    gradient and to its norm (comat_adamw's grad_scale) - no extra pass over the 100 MB buffer."""

    def __init__(self):
        self.pending = []

    def start(self, *flats):
        """launch SUM all-reduces (async); call finish() before the optimizer reads the buffers."""
        if world_size() == 1:
            return
        for f in flats:
            self.pending.append((dist.all_reduce(f, op=dist.ReduceOp.SUM, async_op=True), f))

    def finish(self):
        """wait for the collectives; -> 1 / world (the scale that turns the summed buffers into means)"""
        for work, _ in self.pending:
            work.wait()
        self.pending = []
        return 1.0 / world_size()


def barrier():
    if dist.is_initialized():
        dist.barrier()


def all_gather_scalar(x: float, device):
    if not dist.is_initialized():
        return [x]
    t = torch.tensor([x], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o) for o in out]
