"""Frame sharding across the GPUs of one node (one process per GPU).

The hot path shards over independent frames: separate ``fiasco_coder``
calls / grayscale all-intra frames share no state, so rank r simply encodes frames
r, r+W, r+2W, ... with no data-path collective.  The only exchange is the trivial
gather of the finished byte strings (a few KB per frame): one all-reduce of the lengths
and one padded all-gather of the payloads.  With backend "nccl" this is RCCL over xGMI on
device tensors; with "gloo" the same code runs on CPU tensors (tests).
"""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """Round-robin assignment of work items to ranks."""
    return list(range(rank, n_items, world))


def gather_streams(local, n_items, device="cpu", group=None):
    """local: {global index: bytes} of this rank.  Returns the list of all n_items byte
    strings in global order on every rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local[i] for i in range(n_items)]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lengths = torch.zeros(n_items, dtype=torch.int64, device=device)
    for i, b in local.items():
        lengths[i] = len(b)
    dist.all_reduce(lengths, op=dist.ReduceOp.SUM, group=group)
    per_rank = (n_items + world - 1) // world
    maxlen = int(lengths.max().item()) if n_items else 0
    mine = torch.zeros((per_rank, max(maxlen, 1)), dtype=torch.uint8, device=device)
    for k, i in enumerate(shard_indices(n_items, rank, world)):
        b = local[i]
        mine[k, :len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = [None] * n_items
    lens = lengths.cpu().tolist()
    for r in range(world):
        p = parts[r].cpu()
        for k, i in enumerate(shard_indices(n_items, r, world)):
            out[i] = bytes(p[k, :lens[i]].numpy().tobytes())
    return out
