"""Data-parallel inference across the GPUs of one node.

The reference has no multi-GPU inference (production.py is one process, one GPU, one image at a
time).  The path shards embarrassingly: images are independent, every rank holds the full model
weights and the full gallery matrix.  The only collective is at start-up: the gallery is EMBEDDED
sharded (each rank G/world rows) and assembled with one all_gather over RCCL/xGMI (6.6-13 MB for
G = 3.2k) -- steady state has no collectives at all.
"""
import os

import torch
import torch.distributed as dist


# every collective this process issues, by kind: bench.py prints it (`config.collectives`) so that a run on N GPUs shows that the
# process group really had N ranks, that the gallery all_gather ran once, and that no collective sits inside the timed steps
STATS = {'all_gather': 0, 'all_gather_bytes_sent': 0, 'all_gather_bytes_received': 0, 'all_reduce': 0, 'barrier': 0}


def collective_stats():
    """-> dict | None (no process group): backend, the world size THE PROCESS GROUP reports, and the counts / bytes above."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dict(STATS, backend=dist.get_backend(), world_size=dist.get_world_size())


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def force_collectives():
    """CVPCE_DIST_FORCE_COLLECTIVES=1: a 1-rank job still creates its process group and runs every collective of the N-rank
    path (RCCL smoke test on a one-GPU box: init, all_gather and all_reduce on device tensors, barrier)."""
    return os.environ.get('CVPCE_DIST_FORCE_COLLECTIVES', '0') == '1'


def init(backend=None):
    """One process per GPU (torchrun env).  backend 'nccl' IS RCCL on ROCm; 'gloo' for CPU tests."""
    rank, local_rank, world = env_world()
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('CVPCE_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n, rank, world):
    """Contiguous block partition of n units: the first n % world ranks get one extra."""
    q, r = divmod(n, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_images(num_images, rank, world):
    """Image i of a global batch goes to the rank owning its contiguous block (8e)."""
    return list(range(*shard_range(num_images, rank, world)))


def all_gather_rows(local, total_rows, rank, world):
    """Assemble a (total_rows, D) matrix from per-rank row blocks laid out by shard_range."""
    if world == 1 and not (force_collectives() and dist.is_initialized()):
        return local
    sizes = [shard_range(total_rows, r, world) for r in range(world)]
    max_rows = max(e - s for s, e in sizes)
    dev = local.device
    stage = torch.device('cpu') if dist.get_backend() == 'gloo' else dev   # gloo rehearsals stage through host memory
    pad = torch.zeros((max_rows, local.shape[1]), dtype=local.dtype, device=stage)
    pad[:local.shape[0]] = local.to(stage)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    STATS['all_gather'] += 1
    STATS['all_gather_bytes_sent'] += pad.numel() * pad.element_size()
    STATS['all_gather_bytes_received'] += pad.numel() * pad.element_size() * world
    return torch.cat([o[:e - s] for o, (s, e) in zip(out, sizes)]).to(dev)


def build_gallery_sharded(embed_fn, gallery_images, rank, world):
    """Each rank embeds its block of gallery images with `embed_fn` ((b,3,256,256) -> (b,D)); one all_gather."""
    s, e = shard_range(len(gallery_images), rank, world)
    local = embed_fn(gallery_images[s:e])
    return all_gather_rows(local, len(gallery_images), rank, world)


def max_over_ranks(value, device):
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device='cpu' if dist.get_backend() == 'gloo' else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    STATS['all_reduce'] += 1
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        STATS['barrier'] += 1
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])     # RCCL: the barrier's all_reduce runs on THIS rank's GPU
        else:
            dist.barrier()


def backend_name():
    """'nccl' (= RCCL) | 'gloo' | None when this job runs without a process group."""
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None
