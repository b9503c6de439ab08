"""Frame and GOP sharding across the GPUs of one node (one process per GPU).

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


def encode_sequence(lib, pnm_list, quality=20.0, options=None, device="cpu", group=None):
    """One video over all ranks: rank r searches and writes the groups of pictures r, r+W, ...
    (independent but for two things a colour stream carries from frame to frame):

      * the minimum block level (reference codec/coder.c:785-797) -- every GOP starts from a
        SPECULATED level; one all-reduce gathers what each GOP left; the chain is verified on
        every rank alike and the GOPs behind the first wrong start value are searched again;
      * the y_column flags (codec/wfalib.c:277-310) -- one all-reduce gathers the raw flags of
        all frames, every rank resolves them in coding order and writes its own frames.

    The byte strings of the frames are then gathered like independent frames (gather_streams) and
    put together in coding order.  Returns the .fco bytes on every rank."""
    import fiasco_amd
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    seq = fiasco_amd.Sequence(lib, pnm_list, quality, options, rank, world)
    # An error on ONE rank (out of HBM, a HIP error on its GPU ...) must not leave the others
    # blocked in the next collective: local errors are caught, folded into the tensor that is
    # reduced anyway (one extra row / element), and every rank raises together after the reduce.
    local_err = None

    def _raise_together(flag_sum, what):
        if flag_sum:
            raise fiasco_amd.FiascoError(local_err or "%s failed on another rank" % what)

    try:
        G, K = seq.gops, seq.frames
        # every rank probes the first frame itself: the same guess everywhere, nothing to send
        try:
            guess = seq.probe()
        except fiasco_amd.FiascoError as e:
            local_err, guess = str(e), seq.initial_level
        carry = [seq.initial_level] + [guess] * (G - 1)
        todo = [True] * G
        used, left, failed = [0] * G, [0] * G, [False] * G
        for _ in range(64):
            mine = torch.zeros((G + 1, 3), dtype=torch.int64, device=device)
            if local_err is None:
                try:
                    seq.search(carry, todo)
                    for g in range(G):
                        if g % world == rank and todo[g]:
                            out, bad, _msg = seq.gop_result(g)
                            mine[g] = torch.tensor([carry[g], out, 1 if bad else 0], dtype=torch.int64, device=device)
                except fiasco_amd.FiascoError as e:
                    local_err = str(e)
            if local_err is not None:
                mine[G, 0] = 1
            if world > 1:
                dist.all_reduce(mine, op=dist.ReduceOp.SUM, group=group)
            got = mine.cpu().tolist()
            _raise_together(got[G][0], "the search of a group of pictures")
            for g in range(G):
                if todo[g]:
                    used[g], left[g], failed[g] = got[g][0], got[g][1], bool(got[g][2])
            t, first_invalid = seq.initial_level, G
            for g in range(G):
                if used[g] != t:
                    first_invalid = g
                    break
                if failed[g]:
                    raise fiasco_amd.FiascoError("group of pictures %d: the coder failed" % g)
                t = left[g]
            if first_invalid == G:
                break
            todo = [g >= first_invalid for g in range(G)]
            carry = [t if todo[g] else carry[g] for g in range(G)]
        else:
            raise fiasco_amd.FiascoError("minimum level chain does not settle")
        # y_column chain (colour only): raw flags of every frame -> resolved flags of my frames
        resolved = {}
        if seq.ycol_size:
            raw = torch.zeros((K, seq.ycol_size), dtype=torch.uint8, device=device)
            for k in range(K):
                if seq.gop_of(k) % world == rank:
                    raw[k] = torch.frombuffer(bytearray(seq.ycol(k)), dtype=torch.uint8).to(device)
            if world > 1:
                dist.all_reduce(raw, op=dist.ReduceOp.SUM, group=group)
            raw = raw.cpu()
            chain = torch.zeros(seq.ycol_size, dtype=torch.uint8)
            for k in range(K):
                chain = torch.where(raw[k] != 2, raw[k], chain)
                if seq.gop_of(k) % world == rank:
                    resolved[k] = bytes(chain.numpy().tobytes())
        local = {}
        try:
            local = {k: seq.write(k, resolved.get(k)) for k in range(K) if seq.gop_of(k) % world == rank}
        except fiasco_amd.FiascoError as e:
            local_err = str(e)
    finally:
        seq.free()
    if world == 1:
        _raise_together(local_err is not None, "the stream writer")
        return b"".join(local[k] for k in range(K))
    # gather_streams expects item i on rank i % world: the frames are owned by GOP, so gather in
    # two steps -- lengths (+ one element for "a rank failed"), then the payloads padded to the longest
    lengths = torch.zeros(K + 1, dtype=torch.int64, device=device)
    if local_err is not None:
        lengths[K] = 1
    else:
        for k, b in local.items():
            lengths[k] = len(b)
    dist.all_reduce(lengths, op=dist.ReduceOp.SUM, group=group)
    lens = lengths.cpu().tolist()
    _raise_together(lens[K], "the stream writer")
    lens = lens[:K]
    buf = torch.zeros((K, max(lens) if K else 1), dtype=torch.uint8, device=device)
    for k, b in local.items():
        buf[k, :len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    buf = buf.cpu()
    return b"".join(bytes(buf[k, :lens[k]].numpy().tobytes()) for k in range(K))
