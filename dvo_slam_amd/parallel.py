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


class PendingGather:
    """An all-gather in flight (gather_records_start); result() waits for it and restores global pair order."""

    def __init__(self, n_pairs, world_size, work, out, keep):
        self.n_pairs, self.world_size, self.work, self.out, self.keep = n_pairs, world_size, work, out, keep

    def result(self):
        if self.work is not None:
            self.work.wait()
        full = np.zeros((self.n_pairs, RECORD), np.float64)
        for r in range(self.world_size):
            idx = shard_indices(self.n_pairs, r, self.world_size)
            full[idx] = self.out[r][: len(idx)].cpu().numpy()
        return full


def gather_records_start(local_records, n_pairs, rank, world_size, device=None):
    """Start the all-gather of the per-rank record blocks (asynchronous: the collective of batch k travels while batch k+1
    is being aligned).  Needs an initialised torch.distributed process group (backend "nccl" = RCCL on the GPUs, "gloo" in
    CPU tests).  Ranks may own different numbers of pairs (n_pairs need not divide by world_size): blocks are padded."""
    import torch
    import torch.distributed as dist
    per_rank = (n_pairs + world_size - 1) // world_size
    buf = torch.zeros((per_rank, RECORD), dtype=torch.float64, device=device)
    loc = torch.from_numpy(np.ascontiguousarray(local_records, dtype=np.float64).reshape(-1, RECORD))
    buf[: loc.shape[0]] = loc.to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world_size)]
    work = dist.all_gather(out, buf, async_op=True)
    return PendingGather(n_pairs, world_size, work, out, (buf, loc))


def gather_records(local_records, n_pairs, rank, world_size, device=None):
    """All-gather the per-rank record blocks and restore global pair order (blocking)."""
    if world_size == 1:
        return np.asarray(local_records, np.float64).reshape(-1, RECORD)
    return gather_records_start(local_records, n_pairs, rank, world_size, device).result()
