"""N>1 path on CPU: 2 processes, gloo backend.  Each rank encodes its round-robin shard of
a list of independent frames (here with the CPU oracle, because this container has no GPU)
and the byte strings are gathered exactly as bench.py does with RCCL on the GPU box."""
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, paths, ref, oracle_lib):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import fiasco_amd
    from fiasco_amd.sharding import shard_indices, gather_streams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = fiasco_amd.Library(oracle_lib)
    lib.set_verbosity(0)
    o = lib.cli_options()
    frames = [open(p, "rb").read() for p in paths]
    mine = shard_indices(len(frames), rank, world)
    enc = lib.encode_batch([frames[i] for i in mine], 20.0, o)
    local = dict(zip(mine, enc))
    allb = gather_streams(local, len(frames), device="cpu")
    dist.barrier()
    assert allb == ref, "rank %d: gathered streams differ" % rank
    dist.destroy_process_group()


def test_shard_indices_cover_everything():
    from fiasco_amd.sharding import shard_indices
    for n in (0, 1, 5, 8, 64):
        for w in (1, 2, 4, 8):
            got = sorted(i for r in range(w) for i in shard_indices(n, r, w))
            assert got == list(range(n))


def test_two_rank_gloo_shard_and_gather(oracle, inputs, tmp_path):
    names = ["g96x64", "g64x32", "g32x32", "n128x96", "f0_96x64"]      # ragged sizes, odd count
    paths = [inputs.path(n) for n in names]
    o = oracle.cli_options()
    ref = oracle.encode_batch([open(p, "rb").read() for p in paths], 20.0, o)
    assert all(r is not None for r in ref)
    from conftest import ORACLE_LIB
    mp.spawn(_worker, args=(2, 29531, paths, ref, ORACLE_LIB), nprocs=2, join=True)


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun environment) must become a 2-rank job and print
    ONE line with n_gpus == 2.  FIASCO_BENCH_DRYRUN=1 takes the device coder out (this container
    has no GPU): launcher, rendezvous on 127.0.0.1, max-over-ranks timing, the reductions and the
    gather of the per-rank streams run exactly as on the GPU box, over gloo."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FIASCO_BENCH_DRYRUN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "0", "--frames-per-gpu", "6"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["dry_run"] is True and j["scaling"] == "weak"
    assert j["config"]["parallelism"] == "frames x2"
