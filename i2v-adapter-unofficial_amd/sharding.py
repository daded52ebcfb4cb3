"""Multi-GPU layout of the denoising path: independent (image, prompt) samples shard across ranks,
one process per GPU; the only collective is the start-up broadcast of the shared fp16 weights from rank 0 as ONE
flat buffer in state-dict order (RCCL over xGMI on GPUs, gloo on CPU for the tests).  There is no per-step
collective: every sample's trajectory is independent (no cross-sample op anywhere in unet:1289-1451).
"""
from typing import Dict, List, Sequence, Tuple

import torch


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Static block partition [begin, end) of n_items over world_size ranks (first `n_items % world_size` ranks
    get one extra item)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} / world size {world_size}")
    base, extra = divmod(n_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_items(items: Sequence, rank: int, world_size: int) -> List:
    b, e = shard_range(len(items), rank, world_size)
    return list(items[b:e])


def flatten_state(state: Dict[str, torch.Tensor], dtype=torch.float16) -> Tuple[torch.Tensor, List[Tuple[str, torch.Size, int]]]:
    """One contiguous buffer holding every tensor of `state` (state-dict order) + the index to unpack it."""
    index, total = [], 0
    for k, v in state.items():
        index.append((k, v.shape, total))
        total += v.numel()
    dev = next(iter(state.values())).device
    flat = torch.empty(total, dtype=dtype, device=dev)
    for (k, shape, off), v in zip(index, state.values()):
        flat[off: off + v.numel()].copy_(v.reshape(-1))
    return flat, index


def broadcast_model_weights(model: torch.nn.Module, src: int = 0, group=None) -> int:
    """Broadcast every parameter and buffer of `model` from rank `src` as a single flat buffer.
    All ranks must hold a model of the same architecture, dtype and device.  Returns the number of bytes sent."""
    import torch.distributed as dist
    state = model.state_dict()
    flat, index = flatten_state(state, dtype=next(iter(state.values())).dtype)
    dist.broadcast(flat, src=src, group=group)
    if dist.get_rank(group) != src:
        with torch.no_grad():
            for (k, shape, off), v in zip(index, state.values()):
                v.copy_(flat[off: off + v.numel()].view(shape))
    return flat.numel() * flat.element_size()


def gather_latents(latents: torch.Tensor, dst: int = 0, group=None):
    """Collect each rank's final latents on rank `dst` (report only; 0.5 MB per sample at 512^2)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_rank(group) == dst:
        out = [torch.empty_like(latents) for _ in range(world)]
        dist.gather(latents, out, dst=dst, group=group)
        return out
    dist.gather(latents, None, dst=dst, group=group)
    return None
