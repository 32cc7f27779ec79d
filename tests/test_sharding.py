"""N>1 path on CPU: two gloo processes shard a datalist round-robin and gather variable-length
payloads to rank 0 in datalist order (the only exchange step of the multi-GPU design)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _payload(i):
    return bytes([(i * 7 + k) % 251 for k in range(10 + (i * 37) % 90)])


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    import rpcc_amd  # noqa: F401
    from rpcc_amd.sharding import gather_payloads, shard_indices
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_items, rank, world)
    out = gather_payloads([_payload(i) for i in mine], torch.device("cpu"))
    if rank == 0:
        q.put([o == _payload(i) for i, o in enumerate(out)])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_two_rank_gloo_gather(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + n_items
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res) == n_items and all(res)


def test_shard_indices_partition():
    sys.path.insert(0, ROOT)
    import rpcc_amd  # noqa: F401
    from rpcc_amd.sharding import owner_of, shard_indices
    for n in (0, 1, 5, 16, 17):
        for w in (1, 2, 3, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert all(owner_of(i, w) == r for r in range(w) for i in parts[r])
