"""N>1 data-parallel path on CPU: 2 ranks over gloo, kernels simulated (tests/sim_backend.py).  Checks the DDP
semantics the reference gets from accelerate (training_script.py:659): after the step every rank holds the MEAN of
the ranks' LoRA gradients and identical updated parameters, although each rank saw a different prompt/latents."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, gpu=False, segments=False):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.set_num_threads(2)
    from comat_amd import dist as cdist
    from comat_amd import ops
    from test_step import make_world
    if gpu:  # the real kernels, one process per GPU, RCCL ("nccl" backend on ROCm) over xGMI
        from comat_amd import _hip
        ops.set_kernel_backend(_hip.HipKernels())
        r, w, dev = cdist.init(backend="nccl")
        assert dev.type == "cuda"
    else:
        from sim_backend import SimKernels
        ops.set_kernel_backend(SimKernels())
        r, w, dev = cdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    cfg, batch, W, trainer = make_world(torch.float32, dev, False)
    g = torch.Generator().manual_seed(100 + rank)  # each rank: its own prompt / latents
    batch["latents"] = torch.randn(batch["latents"].shape, generator=g)
    batch["prompt_embeds"] = torch.randn(batch["prompt_embeds"].shape, generator=g)
    # local gradient of this rank (no reduction), for the reference mean
    trainer.bank.zero_grad()
    out = trainer.compute_losses(batch, training_steps=[1, 2], crop=(0, 0, 63, 63))
    out["loss"].backward()
    local = trainer.bank.flat_grad.clone()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = torch.stack(gathered).mean(0)
    # the real step: reduces, clips, updates
    p0 = trainer.bank.flat.clone()
    if segments:  # the launch path bench.py uses on 1 and on N GPUs alike (dry on CPU: the segments run eagerly)
        from comat_amd.segments import SegmentedStep
        SegmentedStep(trainer, dry=not gpu)(batch, training_steps=[1, 2], crop=(0, 0, 63, 63))
    else:
        trainer.train_step(batch, training_steps=[1, 2], crop=(0, 0, 63, 63))
    cpu = lambda t: t.detach().cpu().clone()
    # the buffer holds the ranks' SUM after the exchange; the optimizer kernel applies 1 / world (grad_scale)
    torch.save(dict(mean=cpu(mean), reduced=cpu(trainer.bank.flat_grad) / world, params=cpu(trainer.bank.flat), p0=cpu(p0),
                    local=cpu(local), d_params=cpu(trainer.D.bank.flat)), os.path.join(out_dir, f"rank{rank}.pt"))
    cdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("segments", [False, True])
def test_two_rank_gloo_grad_mean(tmp_path, segments):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), False, segments), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt")) for i in range(world)]
    assert not torch.allclose(r[0]["local"], r[1]["local"])           # ranks really saw different data
    for i in range(world):
        assert torch.allclose(r[i]["reduced"], r[i]["mean"], rtol=1e-5, atol=1e-7)  # all-reduce(mean)
    assert torch.equal(r[0]["reduced"], r[1]["reduced"])
    assert torch.equal(r[0]["params"], r[1]["params"])                  # replicas stay in sync
    assert not torch.equal(r[0]["params"], r[0]["p0"])
    # ... and the update is clip + AdamW on the MEAN gradient (the 1 / world factor lives inside the optimizer kernel)
    from comat_amd.step import StepConfig
    c = StepConfig()
    p = r[0]["p0"].clone().requires_grad_(True)
    p.grad = r[0]["mean"].clone()
    opt = torch.optim.AdamW([p], lr=1e-2, betas=(c.adam_beta1, c.adam_beta2), eps=c.adam_epsilon, weight_decay=c.adam_weight_decay)
    torch.nn.utils.clip_grad_norm_([p], c.max_grad_norm)
    opt.step()
    assert torch.allclose(p.detach(), r[0]["params"], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_rank_rccl_grad_mean(tmp_path):
    """The same check on two real GPUs over RCCL (one process per GPU): all-reduced gradient == mean of the ranks' local
    gradients, generator and discriminator replicas bit-identical after the step (the G all-reduce is launched while the
    D step still runs on its own stream).  Skips on a 1-GPU box (gpurun boxes have one GPU; the driver's 8-GPU node runs
    it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, True), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt")) for i in range(world)]
    assert not torch.allclose(r[0]["local"], r[1]["local"])
    for i in range(world):
        assert torch.allclose(r[i]["reduced"], r[i]["mean"], rtol=1e-5, atol=1e-7)
    assert torch.equal(r[0]["reduced"], r[1]["reduced"])
    assert torch.equal(r[0]["params"], r[1]["params"]) and torch.equal(r[0]["d_params"], r[1]["d_params"])
    assert not torch.equal(r[0]["params"], r[0]["p0"])
