"""CPU tier: the N>1 path (pair sharding + result all-gather) with world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from dvo_slam_amd import parallel as par


def test_shard_and_records_roundtrip():
    assert par.shard_indices(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((par.shard_indices(1024, r, 8) for r in range(8)), [])) == list(range(1024))
    rng = np.random.default_rng(0)
    tw = rng.normal(size=(5, 6))
    M = rng.normal(size=(5, 6, 6))
    info = M + M.transpose(0, 2, 1)
    ll = rng.normal(size=5)
    t2, i2, l2, f2 = par.unpack_records(par.pack_records(tw, info, ll, np.arange(5)))
    assert np.array_equal(t2, tw) and np.allclose(i2, info) and np.array_equal(l2, ll) and np.array_equal(f2, np.arange(5))
    assert np.array_equal(par.gather_records(par.pack_records(tw, info, ll), 5, 0, 1), par.pack_records(tw, info, ll))


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = par.shard_indices(n_pairs, rank, world)
    # stand-in for the per-rank GPU alignment: a deterministic function of the global pair index
    tw = np.array([[i + 0.1 * k for k in range(6)] for i in idx], dtype=np.float64).reshape(-1, 6)
    info = np.array([np.eye(6) * (i + 1) for i in idx]).reshape(-1, 6, 6)
    rec = par.pack_records(tw, info, [float(i) for i in idx], [rank] * len(idx))
    full = par.gather_records(rec, n_pairs, rank, world)
    # the pipelined form bench.py uses: the gather of batch k is collected after batch k+1 has been started
    first = par.gather_records_start(rec, n_pairs, rank, world)
    second = par.gather_records_start(rec * 2.0, n_pairs, rank, world)
    assert np.array_equal(first.result(), full) and np.array_equal(second.result(), full * 2.0)
    dist.barrier()
    q.put((rank, full))
    dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n_pairs = 7   # ragged: ranks own 4 and 3 pairs
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1])
    tw, info, ll, flag = par.unpack_records(got[0])
    for i in range(n_pairs):
        assert np.allclose(tw[i], [i + 0.1 * k for k in range(6)])
        assert np.allclose(info[i], np.eye(6) * (i + 1))
        assert ll[i] == i and flag[i] == i % 2
