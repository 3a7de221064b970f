"""Sharding of independent frame pairs over the GPUs of one node.

The alignment path shards embarrassingly by frame pair -- the reference already runs independent match()
calls on a TBB thread pool (dvo_slam/src/keyframe_graph.cpp:576-593, dvo_slam/src/local_tracker.cpp:180-184).
One process per GPU; pair i goes to rank i mod G; there is no communication while aligning.  The only exchange
is one all-gather of fixed-size result records (twist[6] + upper-triangular information[21] + loglik + flags = 32
doubles = 256 B per pair) per batch -- RCCL over xGMI when the process group's backend is "nccl", latency-bound.
"""
import numpy as np

RECORD = 32   # doubles per pair


def shard_indices(n_pairs, rank, world_size):
    """Indices of the pairs rank `rank` aligns (round-robin)."""
    return list(range(rank, n_pairs, world_size))


def pack_records(twists, informations, logliks, flags=None):
    """-> float64 array [n, RECORD]: twist(6) | information upper triangle(21) | loglik | flag | pad(3)"""
    n = len(twists)
    rec = np.zeros((n, RECORD), np.float64)
    iu = np.triu_indices(6)
    for i in range(n):
        rec[i, 0:6] = twists[i]
        rec[i, 6:27] = np.asarray(informations[i])[iu]
        rec[i, 27] = logliks[i]
        rec[i, 28] = 0.0 if flags is None else flags[i]
    return rec


def unpack_records(rec):
    iu = np.triu_indices(6)
    twists = rec[:, 0:6].copy()
    infos = np.zeros((rec.shape[0], 6, 6))
    for i in range(rec.shape[0]):
        infos[i][iu] = rec[i, 6:27]
        infos[i] = infos[i] + infos[i].T - np.diag(np.diag(infos[i]))
    return twists, infos, rec[:, 27].copy(), rec[:, 28].copy()


def gather_records(local_records, n_pairs, rank, world_size, device=None):
    """All-gather the per-rank record blocks and restore global pair order.  Needs an initialised
    torch.distributed process group when world_size > 1 (backend "nccl" = RCCL on the GPUs, "gloo" in CPU tests).
    Ranks may own different numbers of pairs (n_pairs need not divide by world_size): blocks are padded."""
    if world_size == 1:
        return np.asarray(local_records, np.float64).reshape(-1, RECORD)
    import torch
    import torch.distributed as dist
    per_rank = (n_pairs + world_size - 1) // world_size
    buf = torch.zeros((per_rank, RECORD), dtype=torch.float64, device=device)
    loc = torch.from_numpy(np.ascontiguousarray(local_records, dtype=np.float64).reshape(-1, RECORD))
    buf[: loc.shape[0]] = loc.to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world_size)]
    dist.all_gather(out, buf)
    full = np.zeros((n_pairs, RECORD), np.float64)
    for r in range(world_size):
        idx = shard_indices(n_pairs, r, world_size)
        full[idx] = out[r][: len(idx)].cpu().numpy()
    return full
