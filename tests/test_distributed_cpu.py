"""The N>1 path on CPU: world_size-2 gloo job running the sharded harness with a stand-in
model (the real engines need an MI355X).  Checks the reference's chunk rule
(generate.py:204: batch i -> rank i % world; :304: seed + 1000*rank) and the gather."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import REPO


class FakeModel:
    """Deterministic stand-in with the wrapper API; codes depend on (conditioning, torch RNG)."""
    codes_size, image_size = 4, 8
    device = torch.device("cpu")

    def sample(self, conditioning, gen_params, apply_watermark=False):
        c = torch.as_tensor(conditioning).view(-1, 1)
        noise = torch.randint(0, 1000, (c.shape[0], 16))
        return (c * 7 + noise) % 512

    def codes_to_images(self, codes):
        return (codes.float().view(-1, 1, 4, 4).repeat(1, 3, 2, 2) / 256.0 - 1.0).clamp(-1, 1)

    def images_to_codes(self, images):
        return ((images[:, 0, :4, :4] + 1.0) * 256.0).round().long().view(-1, 16)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wmar_amd import harness
    inputs = [c for c in (1, 9, 232) for _ in range(5)]  # 15 images, batch 4 -> 4 batches
    ev = {"metric_names": ["l0"], "augmentations": [], "max_roundtrips": 1, "orig_only": False}
    recs = harness.generate_sharded(None, FakeModel(), inputs, None, ev, {"batch_size": 4}, seed=1)
    if rank == 0:
        q.put(recs)
    dist.barrier()
    dist.destroy_process_group()


def _single(chunk_id, num_chunks):
    from wmar_amd import harness
    harness.seed_everything(1, chunk_id)
    inputs = [c for c in (1, 9, 232) for _ in range(5)]
    ev = {"metric_names": ["l0"], "augmentations": [], "max_roundtrips": 1, "orig_only": False}
    return harness.generate(None, FakeModel(), inputs, None, ev, {"batch_size": 4}, chunk_id=chunk_id,
                            num_chunks=num_chunks)


@pytest.mark.timeout(120)
def test_two_rank_job_equals_two_reference_chunks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    recs = q.get(timeout=90)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # the same work as two independent "chunk" processes of the reference's job array
    ref = _single(0, 2) + _single(1, 2)
    ref.sort(key=lambda r: (r["batch_idx"], r["idx"], r["transform"], str(r["param"])))
    strip = lambda rs: [(r["batch_idx"], r["conditioning"], r["idx"], r["transform"], r["param"], r["metrics"]) for r in rs]
    assert strip(recs) == strip(ref)
    # every image exactly once, per-conditioning indices advance across skipped batches (generate.py:193-207)
    seen = sorted((r["conditioning"], r["idx"]) for r in recs if r["transform"] == "roundtrips" and r["param"] == 0)
    assert seen == sorted((c, i) for c in (1, 9, 232) for i in range(1, 6))
    assert {r["batch_idx"] % 2 for r in recs} == {0, 1}


def test_make_batches_and_seed_rule():
    from wmar_amd import harness
    assert harness.make_batches(list(range(10)), 4) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]
    assert harness.make_batches([], 4) == []
    assert harness.seed_everything(42, 3) == 3042


# ------------------------------------------------------------------ BASELINE configs[4]: 512 conditionings over 8 ranks
def _worker8(rank, world, port, q):
    import hashlib

    import numpy as np
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wmar_amd import harness
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    # the key: built on rank 0 only, broadcast, and compared on every rank with a local build (bit-equal)
    vq = {"alive_ids": torch.arange(512), "dead_ids": torch.zeros(0, dtype=torch.long), "embedding": None}
    wm = GentimeWatermark(vq, 512, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cpu")
    table = harness.broadcast_key_table(wm, torch.device("cpu"))
    got = hashlib.sha256(table.numpy().tobytes()).hexdigest()
    own = hashlib.sha256(wm.key_table_host().view(np.int32).tobytes()).hexdigest()
    inputs = [(i * 37) % 1000 for i in range(512)]          # bench.py's conditioning rule; 512 images, batch 64 -> one batch per rank
    ev = {"metric_names": ["l0"], "augmentations": [], "max_roundtrips": 1, "orig_only": False}
    recs = harness.generate_sharded(None, FakeModel(), inputs, None, ev, {"batch_size": 64}, seed=1)
    hashes = [None] * world
    dist.all_gather_object(hashes, (got, own))
    if rank == 0:
        q.put((recs, hashes))
    dist.barrier()
    dist.destroy_process_group()


def _single8(chunk_id):
    from wmar_amd import harness
    harness.seed_everything(1, chunk_id)
    inputs = [(i * 37) % 1000 for i in range(512)]
    ev = {"metric_names": ["l0"], "augmentations": [], "max_roundtrips": 1, "orig_only": False}
    return harness.generate(None, FakeModel(), inputs, None, ev, {"batch_size": 64}, chunk_id=chunk_id, num_chunks=8)


@pytest.mark.timeout(300)
def test_eight_rank_job_equals_eight_reference_chunks_and_one_key():
    """BASELINE.json configs[4]: 512 conditionings sharded over 8 ranks (64 per rank and step) -- the job equals the reference's
    eight-chunk job array (generate.py:204 batch striping, :304 seed + 1000 * chunk), every image exactly once, and the key table
    built on rank 0 arrives bit-equal on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    recs, hashes = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({h for pair in hashes for h in pair}) == 1, hashes
    ref = sum((_single8(c) for c in range(8)), [])
    ref.sort(key=lambda r: (r["batch_idx"], r["idx"], r["transform"], str(r["param"])))
    strip = lambda rs: [(r["batch_idx"], r["conditioning"], r["idx"], r["transform"], r["param"], r["metrics"], r["codes"].tolist()) for r in rs]
    assert strip(recs) == strip(ref)
    base = [r for r in recs if r["transform"] == "roundtrips" and r["param"] == 0]
    assert len(base) == 512 and sorted({r["batch_idx"] for r in base}) == list(range(8))
    assert all(sum(1 for r in base if r["batch_idx"] == b) == 64 for b in range(8))
