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


def test_vectorised_twists_match_the_oracle_log():
    from oracle import pyoracle as po
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.normal(scale=0.3, size=(40, 6)), rng.normal(scale=1e-9, size=(4, 6)), np.zeros((1, 6))])
    Ts = np.stack([po.se3_exp(x) for x in xs])
    assert np.abs(par.twists_of(Ts) - xs).max() < 1e-12


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
    # the pipelined form bench.py uses: buffers allocated once, the gather of batch k collected after batch k+1 has been started
    g = par.RecordGatherer(n_pairs, rank, world)
    first = g.start(rec)
    second = g.start(rec * 2.0)
    assert np.array_equal(first.result(), full) and np.array_equal(second.result(), full * 2.0)
    third = g.start(rec * 3.0)                       # the first slot again
    assert np.array_equal(third.result(), full * 3.0)
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


# ---- GPU tier: real device records through the N > 1 path -----------------------------------------------------------
import json      # noqa: E402
import subprocess  # noqa: E402
import sys       # noqa: E402

import pytest    # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_without_a_launcher_starts_its_ranks():
    """CPU tier: `python bench.py --gpus 2` with WORLD_SIZE unset re-executes itself under torch.distributed.run.  Without a GPU the
    ranks stop at the device check (there is no CPU fallback) -- what matters here is that THEY ran: the message is the ranks', not a
    refusal to start, and the launcher passes their failure on."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU tier runs the real thing (test_bench_launches_its_own_ranks)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert b.returncode != 0
    assert "needs an MI355X" in b.stderr and "launch with torch.distributed.run" not in b.stderr


@pytest.mark.gpu
def test_two_ranks_gather_the_records_of_one_rank(tmp_path):
    """bench.py's N > 1 path with REAL alignments: two processes (sharing GPU 0, gloo -- the box has one GPU; on an 8-GPU node the
    same code runs one rank per GPU over RCCL) align pair i on rank i mod 2 and all-gather the records; the gathered set equals
    the records of a single process that aligns all pairs, bit for bit (tile height and group size pinned)."""
    common = ["--pairs", "12", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-from-host", "--rows-per-wave", "8", "--resident-group", "1"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    a = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--records-out", one] + common,
                       capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                        "--records-out", two] + common, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    ja = json.loads([l for l in a.stdout.splitlines() if l.startswith("{")][-1])
    jb = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
    assert ja["n_gpus"] == 1 and jb["n_gpus"] == 2 and ja["scaling"] == jb["scaling"] == "strong"
    assert ja["config"]["pairs"] == jb["config"]["pairs"] == 12 and jb["config"]["pairs_per_gpu"] == 6
    ra, rb = np.load(one), np.load(two)
    assert ra.shape == rb.shape == (12, par.RECORD)
    assert np.array_equal(ra, rb)
    assert ja["nan_results"] == 0 and ja["max_twist_error_vs_truth"] < 1e-4
    # the library's own RCCL gather asked for where it cannot be had (RCCL refuses two ranks on one device): the ranks AGREE on
    # torch.distributed's all-gather instead of leaving each other inside a collective, and the records are the same
    three = str(tmp_path / "three.npy")
    c = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device",
                        "--gather", "native", "--records-out", three] + common, capture_output=True, text=True, timeout=300)
    assert c.returncode == 0, c.stderr[-2000:]
    assert np.array_equal(np.load(three), ra)


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` started WITHOUT a launcher (the shape of the driver's single-GPU command, WORLD_SIZE unset): bench.py
    becomes the launcher itself -- one rank per GPU under torch.distributed.run -- instead of refusing, exits 0 and rank 0 prints
    exactly one JSON line.  (Two ranks sharing GPU 0 over gloo here; on a multi-GPU node the default backend is RCCL.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = str(tmp_path / "two.npy")
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo", "--pairs", "12",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-from-host", "--records-out", out],
                       capture_output=True, text=True, timeout=900, env=env)
    assert b.returncode == 0, b.stderr[-2000:]
    lines = [l for l in b.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["pairs_per_gpu"] == 6 and j["nan_results"] == 0 and j["value"] > 0
    assert np.load(out).shape == (12, par.RECORD)


@pytest.mark.gpu
def test_rccl_record_path_on_one_gpu(tmp_path):
    """The RCCL code path of the N > 1 record exchange, executed on the one GPU a test box has: bench.py with the gatherer forced on under
    init_process_group("nccl", world_size=1) -- device tensors, pinned staging, the asynchronous all-gather on RCCL's stream, the drain at
    the end of the timed region -- delivers exactly the records a run without any process group packs from its local results.  (What a
    world of one cannot show is the exchange between GPUs over xGMI: the N > 1 curve is the driver's to measure.)"""
    common = ["--pairs", "12", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-from-host", "--no-scaling-model", "--rows-per-wave", "8",
              "--resident-group", "1"]
    plain, forced = str(tmp_path / "plain.npy"), str(tmp_path / "forced.npy")
    a = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--records-out", plain] + common,
                       capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stderr[-2000:]
    ra = np.load(plain)
    # both gatherers: "native" = ncclAllGather called by the C-ABI itself (dvo_hip_gather_records_*, the default with --backend nccl since
    # round 6: what a C++ host uses), "torch" = torch.distributed.all_gather
    for kind in ("native", "torch"):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--backend", "nccl", "--gather", kind,
                            "--records-out", forced] + common, capture_output=True, text=True, timeout=600, env=env)
        assert b.returncode == 0, b.stderr[-2000:]
        jb = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
        assert jb["forced_gather"] == {"backend": "nccl", "world_size": 1, "gather": kind} and jb["n_gpus"] == 1 and jb["nan_results"] == 0
        rb = np.load(forced)
        assert ra.shape == rb.shape == (12, par.RECORD)
        assert np.array_equal(ra, rb), kind


@pytest.mark.gpu
def test_native_gather_through_the_c_abi_world_of_one():
    """dvo_hip_comm_* / dvo_hip_gather_records_* called directly (no torch.distributed anywhere): a communicator of one rank on GPU 0, two
    gathers in flight at once (the double buffering a streaming caller relies on), a smaller share padded with zeros, and the error
    behaviour of the entry points."""
    import ctypes as C
    import dvo_slam_amd as d
    from dvo_slam_amd import _lib
    ctx = d.Context(0)
    L = ctx._lib
    assert L.dvo_hip_context_device(ctx.ptr) == 0
    uid = par.NativeRecordGatherer.unique_id(ctx)
    assert len(uid) == _lib.COMM_ID_BYTES
    g = par.NativeRecordGatherer(ctx, uid, 10, 0, 1)
    assert L.dvo_hip_comm_rank(g.comm) == 0 and L.dvo_hip_comm_size(g.comm) == 1
    rng = np.random.default_rng(3)
    a, b = rng.normal(size=(10, par.RECORD)), rng.normal(size=(7, par.RECORD))
    pa, pb = g.start(a), g.start(b)                                     # both slots in flight
    with pytest.raises(RuntimeError):
        g.start(a)                                                      # a third is refused, not queued
    assert np.array_equal(pa.result(), a)
    got = pb.result()
    assert np.array_equal(got[:7], b) and not got[7:].any()             # the smaller share: padded with zeros
    assert np.array_equal(g.start(b[:3]).result()[:3], b[:3])           # slots are reusable
    t = C.c_int()
    assert L.dvo_hip_gather_records_begin(g.comm, None, 8, 8, C.byref(t)) == _lib.ERR_INVALID
    assert L.dvo_hip_gather_records_begin(g.comm, C.c_void_p(a.ctypes.data), 24, 16, C.byref(t)) == _lib.ERR_INVALID    # more than a block
    assert L.dvo_hip_gather_records_begin(g.comm, C.c_void_p(a.ctypes.data), 4, 12, C.byref(t)) == _lib.ERR_INVALID     # not a multiple of 8
    assert L.dvo_hip_gather_records_end(g.comm, 0, C.c_void_p(a.ctypes.data), 8) == _lib.ERR_INVALID                    # nothing in flight
    g.close()


def test_record_blocks_restore_pair_order():
    """CPU tier: the layout both gatherers deliver -- one block per rank, a rank's pairs in the order it aligned them (pair i on rank
    i mod N as its (i // N)-th), blocks padded to the largest share -- put back into global pair order."""
    for n_pairs, world in ((12, 1), (12, 2), (13, 4), (5, 8), (1024, 8)):
        rec = np.arange(n_pairs * par.RECORD, dtype=np.float64).reshape(n_pairs, par.RECORD) + 1.0
        per_rank = (n_pairs + world - 1) // world
        blocks = np.zeros((world, per_rank, par.RECORD))
        for r in range(world):
            mine = rec[par.shard_indices(n_pairs, r, world)]
            blocks[r, : len(mine)] = mine
        assert np.array_equal(par.blocks_to_pair_order(blocks, n_pairs, world), rec)
        assert np.array_equal(par.blocks_to_pair_order(blocks.reshape(-1), n_pairs, world), rec)


def test_comm_entry_points_refuse_bad_arguments_without_a_device():
    """CPU tier (no compute): the gather entry points of the C-ABI exist and fail cleanly on null arguments."""
    import ctypes as C
    import dvo_slam_amd as d
    from dvo_slam_amd import _lib
    L = d.lib()
    out = C.c_void_p()
    assert L.dvo_hip_comm_create(None, None, 0, 1, C.byref(out)) == _lib.ERR_INVALID and not out.value
    assert L.dvo_hip_comm_get_unique_id(None) == _lib.ERR_INVALID
    assert L.dvo_hip_comm_rank(None) == -1 and L.dvo_hip_comm_size(None) == 0
    t = C.c_int()
    assert L.dvo_hip_gather_records_begin(None, None, 0, 8, C.byref(t)) == _lib.ERR_INVALID
    assert L.dvo_hip_gather_records_end(None, 0, None, 0) == _lib.ERR_INVALID
    L.dvo_hip_comm_destroy(None)
    assert L.dvo_hip_context_device(None) == -1


def test_pipeline_object_packs_the_same_records():
    from oracle import pyoracle as po
    """dvo_stream_pack_records (dvo_slam_amd/apps/stream_pipeline.cpp) -- the per-step record packing of a multi-GPU job done by the
    pipeline object -- against parallel.pack_records(parallel.twists_of(.)): rotations from tiny to near pi, the identity included."""
    import ctypes as C
    from dvo_slam_amd import _lib, stream
    L = stream._load()
    rng = np.random.default_rng(5)
    n = 64
    xi = rng.uniform(-1.0, 1.0, (n, 6))
    xi[0] = 0.0
    xi[1, 3:] *= 1e-9
    xi[2, 3:] *= 3.1 / np.linalg.norm(xi[2, 3:])
    T = np.stack([po.se3_exp(x) for x in xi])
    info = rng.normal(size=(n, 6, 6))
    info = info @ info.transpose(0, 2, 1)
    ll = rng.normal(size=n)
    res = (_lib.Result * n)()
    for i in range(n):
        res[i].transformation[:] = list(T[i].reshape(-1))
        res[i].information[:] = list(info[i].reshape(-1))
        res[i].loglik = ll[i]
    out = np.full((n, par.RECORD), -7.0)
    L.dvo_stream_pack_records(n, res, out.ctypes.data_as(C.POINTER(C.c_double)))
    want = par.pack_records(par.twists_of(T), info, ll)
    assert np.allclose(out, want, rtol=1e-12, atol=1e-13)
    assert np.allclose(out[:, :6], xi, atol=1e-9)
