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


def _seq_worker(rank, world, port, paths, pattern, pred, want, oracle_lib):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import fiasco_amd
    from fiasco_amd.sharding import encode_sequence
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = fiasco_amd.Library(oracle_lib)
    lib.set_verbosity(0)
    o = lib.cli_options(pattern=pattern)
    o.set_prediction(1 if pred else 0, 6, 10)
    got = encode_sequence(lib, [open(p, "rb").read() for p in paths], 20.0, o, device="cpu")
    dist.barrier()
    assert got == want, "rank %d: %d bytes, expected %d" % (rank, len(got), len(want))
    dist.destroy_process_group()


@pytest.mark.parametrize("names,pattern,pred,golden", [
    # colour all-intra: every frame is a GOP, the minimum level ratchets after the first frame --
    # the speculation of the first sweep is WRONG for the later GOPs and they are searched again
    (["c256", "c256b"], "i", 0, "seq2_color_i"),
    (["c256b", "c256"], "i", 0, "seq2_color_i_rev"),
    # y_column flags that show through from frame to frame (found by the fuzzer)
    (["carry48_a", "carry48_b", "carry48_c"], None, 0, "seq3_color_carry48"),
    # P frames: two GOPs of two frames
    (["m0_128x96", "m1_128x96", "m2_128x96", "m3_128x96"], "ipip", 1, None),
    (["c256", "c256b", "c256c", "c256"], "ipip", 0, None),
])
def test_two_rank_gop_sharding_with_speculation(oracle, manifest, inputs, tmp_path, names, pattern, pred, golden):
    """GOPs of one sequence over 2 ranks (gloo): speculate-and-verify of the minimum level chain,
    y_column chain resolved after a gather -- the stream is the one a single fiasco_coder() call
    writes (and, where the real reference produced it, the reference's)."""
    from conftest import ORACLE_LIB, options_from_args
    paths = [inputs.path(n) for n in names]
    if golden:
        case = [c for c in manifest["cases"] if c["name"] == golden][0]
        q, o = options_from_args(oracle, case["args"])
        assert q == 20.0 or golden.startswith("seq3")
    if golden and golden.startswith("seq3"):
        pytest.skip("non-default options: covered through fiasco_coder() in test_oracle_pins")
    o = oracle.cli_options(pattern=pattern or "i")
    o.set_prediction(1 if pred else 0, 6, 10)
    out = str(tmp_path / "single.fco")
    assert oracle.fiasco_coder(paths, out, 20.0, o) == 1, oracle.error_message()
    want = open(out, "rb").read()
    if golden:
        import hashlib
        assert hashlib.md5(want).hexdigest() == case["md5"]        # the real reference's stream
    mp.spawn(_seq_worker, args=(2, 29533, paths, pattern or "i", pred, want, ORACLE_LIB), nprocs=2, join=True)


def _seq_fail_worker(rank, world, port, paths, oracle_lib, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import fiasco_amd
    from fiasco_amd.sharding import encode_sequence
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = fiasco_amd.Library(oracle_lib)
    lib.set_verbosity(0)
    if rank == 1:                      # this rank's device "fails" in its first sweep
        def boom(self, carry_in, todo):
            raise fiasco_amd.FiascoError("out of HBM (simulated)")
        fiasco_amd.Sequence.search = boom
    o = lib.cli_options(pattern="i")
    try:
        encode_sequence(lib, [open(p, "rb").read() for p in paths], 20.0, o, device="cpu")
        q.put((rank, "no error"))
    except fiasco_amd.FiascoError as e:
        q.put((rank, str(e)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_an_error_on_one_rank_raises_on_all_ranks(inputs):
    """A rank whose search fails must not leave the others blocked in the next all-reduce: the
    error is folded into the reduced tensor and every rank raises (fiasco_amd/sharding.py)."""
    from conftest import ORACLE_LIB
    paths = [inputs.path(n) for n in ("c256", "c256b")]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_seq_fail_worker, args=(2, 29537, paths, ORACLE_LIB, q), nprocs=2, join=True)
    got = dict(q.get() for _ in range(2))
    assert "simulated" in got[1] and "another rank" in got[0], got
