"""Frame-level fan-out across GPUs (one process per GPU, torch.distributed).

FFV2 keeps no inter-frame state (reference ffv2enc.c:461-469: CDFs are reset and a
fresh coefficient buffer is built for every frame), so frames are the natural
shard: frame n goes to rank n % world, every rank encodes its own frames with no
data-path collective, and the only exchange is the in-order gather of the
finished packets (a few hundred KB per 4K frame) on rank 0 -- gloo on CPU
tensors in the tests, RCCL over xGMI on the GPU node (backend "nccl").
"""
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def owner(frame: int, world: int) -> int:
    return frame % world


def local_frames(nframes: int, rank: int, world: int) -> List[int]:
    return list(range(rank, nframes, world))


def gather_packets(local: Sequence[bytes], frames: Sequence[int], nframes: int, rank: int, world: int,
                   device: Optional[torch.device] = None, group=None) -> Optional[List[bytes]]:
    """All ranks call this with their own packets (frames[i] -> local[i]).
    Returns the packets in frame order on rank 0, None elsewhere."""
    if world == 1:
        out = [b""] * nframes
        for n, p in zip(frames, local):
            out[n] = p
        return out
    device = device or torch.device("cpu")
    per_rank = (nframes + world - 1) // world
    sizes = torch.zeros(per_rank, dtype=torch.int64, device=device)
    for i, p in enumerate(local):
        sizes[i] = len(p)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    cap = int(max(int(s.max().item()) for s in all_sizes))
    cap = max(cap, 1)
    payload = torch.zeros((per_rank, cap), dtype=torch.uint8, device=device)
    for i, p in enumerate(local):
        payload[i, : len(p)] = torch.frombuffer(bytearray(p), dtype=torch.uint8).to(device)
    bufs = [torch.zeros_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bufs, dst=0, group=group)
    if rank != 0:
        return None
    out: List[bytes] = [b""] * nframes
    for r in range(world):
        host = bufs[r].cpu().numpy()
        for i, n in enumerate(local_frames(nframes, r, world)):
            out[n] = host[i, : int(all_sizes[r][i].item())].tobytes()
    return out


def encode_sequence(encode_batch: Callable[[List[int]], List[bytes]], nframes: int, rank: int, world: int,
                    batch: int = 8, device: Optional[torch.device] = None, group=None) -> Optional[List[bytes]]:
    """encode_batch(list of frame numbers) -> list of packets, called on every rank
    with that rank's frames in chunks of `batch` (frames in flight per GPU)."""
    mine = local_frames(nframes, rank, world)
    packets: List[bytes] = []
    for i in range(0, len(mine), batch):
        packets += encode_batch(mine[i: i + batch])
    return gather_packets(packets, mine, nframes, rank, world, device=device, group=group)
