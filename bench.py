#!/usr/bin/env python
"""bench.py -- frame-pair alignments/s of the MI355X dense RGB-D alignment path (BASELINE.json metric).

Workload = BASELINE config 4: `--pairs` (default 1024) independent 640x480 frame pairs, synthetic seeds 0..pairs-1, 4-level
pyramid, FirstLevel 3, LastLevel 0 (finest level 640x480), MaxIterationsPerLevel 100, Precision 5e-7, Mu 0.  Pair i belongs
to rank i mod N: the total is fixed, every GPU gets pairs/N of it (STRONG scaling); one all-gather of the 256-byte result
records per step over RCCL when N > 1.

A "step" is one pass of the hot path over the rank's whole shard, with the raw sensor planes already resident in HBM:
  re-ingest the raw planes of the shard's frames (u8 grey + u16 depth -> device pyramids, selection)   [SURVEY a11-a14]
  + dvo_hip_match_batch: the coarse-to-fine Gauss-Newton alignment of every pair                      [SURVEY a1-a10]
`value` = pairs aligned per second of that loop with the inputs resident in HBM when the timed region starts (the driver's bench
contract: "inputs already resident in HBM").  SURVEY.md 8d config 4 and BASELINE.md word the same configuration "incl. H2D of 2
planes per frame": that rate -- the same loop fed from pinned HOST memory every step -- is measured right after it in the same run
and reported as `contract_value` / `from_host`, with its own roofline (the PCIe link).  BASELINE config 2 (a single pair) is the same
call with a batch of one and is reported as `single_pair_ms` (latency): one 15 MB pair lives in the 256 MB Infinity Cache and
cannot exercise HBM.

Launch: `python bench.py --gpus 1 --steps K --warmup W`, or for N > 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 640, 480
ALGO_BYTES_PER_PIXEL = 40.0          # SURVEY.md section 8d's ALGORITHMIC bytes: ref {Z,I,Idx,Idy} 16 B + cur {I,Z,Idx,Idy,Zdx,Zdy} 24 B.
                                     # What the shipped sweep MOVES is less (reference {Zsel, I} 8 B + current {I, Z} 8 B x the staged
                                     # window's halo, read; 8 B residual pair written): reported beside it as roofline.moved_bytes_per_pixel
PCIE_PEAK_GBPS = 63.0                # PCIe Gen5 x16, spec (MI355X_MICROARCH.md "Host link")
# (the cap on the workgroups of background build kernels -- library option build_workgroups -- is one per compute unit, the library's policy for
# the device, counter "background_build_workgroups": with the strip ingest the alignment's short kernels get through beside it, r03: 14.23 -> 13.77 ms)
HBM_PEAK_GBPS = 8000.0               # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(pairs_np, cfg_kwargs, sample_pairs, per_thread, trials):
    """The oracle's quirk-faithful REF_SSE mode (kind "port"; bit-identical to the reference's own match(), which is timed beside
    it through oracle/_ref), one match per host thread at a time like the reference's tbb::parallel_reduce over proposals
    (dvo_slam/src/keyframe_graph.cpp:576-593).  Checker code timed as a baseline only.  Every thread count is given `per_thread`
    matches per thread, `trials` times; the median trial counts."""
    from oracle import pyoracle as po
    cores = len(os.sched_getaffinity(0))
    n = min(sample_pairs, pairs_np["grey_ref"].shape[0])
    refs, curs = [], []
    for i in range(n):
        pair = {k: pairs_np[k][i] for k in ("grey_ref", "depth_ref", "grey_cur", "depth_cur")}
        pair["K"] = pairs_np["K"]
        r, c = po.pyramids_from_pair(pair, 4)
        refs.append(r)
        curs.append(c)
    cfg = po.make_config(mode=po.REF_SSE, **cfg_kwargs)
    po.match_batch(refs, curs, cfg, nthreads=min(cores, n))      # warm-up (also builds the per-pyramid caches)

    def rate(threads):
        m = per_thread * threads                                   # pyramids are shared read-only, pair k % n each
        reps = -(-m // n)
        runs = []
        for _ in range(trials):
            _, secs = po.match_batch((refs * reps)[:m], (curs * reps)[:m], cfg, nthreads=threads)
            runs.append(m / secs)
        return float(np.median(runs)), m
    # The path is memory-bound on the CPU too, so more threads are not always faster: a few thread counts, best median reported.
    candidates = sorted({max(1, cores // 8), max(1, cores // 4), max(1, cores // 2), cores})
    table = {t: rate(t) for t in candidates}
    best_threads = max(table, key=lambda t: table[t][0])
    single, m1 = rate(1)
    reference_note, reference_value = "", None
    if po.ref_lib() is not None:
        planes = [(pairs_np["grey_ref"][i].astype(np.float32), po.convert_raw_depth(pairs_np["depth_ref"][i]),
                   pairs_np["grey_cur"][i].astype(np.float32), po.convert_raw_depth(pairs_np["depth_cur"][i])) for i in range(min(n, 16))]
        m = per_thread * best_threads
        runs = []
        for _ in range(trials):
            _, secs = po.ref_match_batch(planes, pairs_np["K"], cfg, n_matches=m, nthreads=best_threads)
            runs.append(m / secs)
        reference_value = float(np.median(runs))
        reference_note = ("; the reference's own DenseTracker::match via oracle/_ref (its translation units compiled against stand-in "
                          "Eigen/OpenCV headers, whose naive matrix algebra makes it slower than the port) on %d threads: %.1f alignments/s"
                          % (best_threads, reference_value))
    return dict(value=table[best_threads][0], unit="alignments/s", cores=best_threads, kind="port", reference_value=reference_value,
                sample="%d matches (%d per thread, median of %d trials) over %d distinct synthetic 640x480 pairs, oracle REF_SSE mode "
                       "(-O3 -march=native), one match per thread at a time; thread counts tried on the %d-thread host: %s; "
                       "single thread: %.1f alignments/s" % (table[best_threads][1], per_thread, trials, n, cores,
                                                             ", ".join("%d -> %.0f/s" % (t, table[t][0]) for t in candidates), single) + reference_note,
                single_thread_value=single, host_threads=cores)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1024, help="GLOBAL number of frame pairs (BASELINE config 4: 1024); every rank aligns pairs/N of them")
    ap.add_argument("--cpu-sample-pairs", type=int, default=32)
    ap.add_argument("--cpu-matches-per-thread", type=int, default=10)
    ap.add_argument("--cpu-trials", type=int, default=3)
    ap.add_argument("--records-out", default="", help="rank 0 writes the gathered result records of the last step here (.npy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--build-workgroups", type=int, default=-1, help="cap on the workgroups of a background build kernel (library option build_workgroups; -1 = bench default)")
    ap.add_argument("--no-overlap", action="store_true", help="build each batch right before its match on one frame set (no build/match overlap)")
    ap.add_argument("--no-from-host", action="store_true", help="skip the PCIe-inclusive leg (raw planes handed over in pinned host memory)")
    ap.add_argument("--no-ref-compat", action="store_true", help="skip the leg that repeats the timed loop with option ref_compat on")
    ap.add_argument("--no-scaling-model", action="store_true", help="skip the one-GPU step times at pairs/2, pairs/4, pairs/8 per step")
    ap.add_argument("--loop-only", action="store_true", help="profiling runs: the timed loop and a minimal line, none of the legs behind it")
    ap.add_argument("--no-guard-stress", action="store_true", help="skip the leg in which 1 %% of the pairs leave the f16 range of the Gram operands")
    ap.add_argument("--rows-per-wave", type=int, default=0)
    ap.add_argument("--resident-group", type=int, default=0, help="workgroups per pair in the resident match kernel (0 = as many as fit); "
                    "with --rows-per-wave: records that do not depend on the batch size")
    ap.add_argument("--lanes", type=int, default=0, help="lanes of the streaming loop on one GPU (contexts + host threads that run their shards' steps out of "
                    "phase, dvo_stream_lanes_*) the TIMED loop runs on: 0 / 1 = one (two lanes are timed as a leg of their own for large batches), N = that many")
    ap.add_argument("--no-lanes-leg", action="store_true", help="skip the leg that times the loop on two lanes (large batches; reported under lanes)")
    ap.add_argument("--lane-depth", type=int, default=2, help="steps the lanes may be submitted ahead of their collection")
    ap.add_argument("--option", action="append", default=[], help="library option key=value (dvo_hip_set_option), for experiments")
    ap.add_argument("--resident-rows", type=int, default=0, help="library option resident_rows (0 = default 24)")
    ap.add_argument("--iters-per-sync", type=int, default=0)
    ap.add_argument("--resident", type=int, default=-1, help="library option resident (-1 default policy, 0 launch path only, 1 every level resident)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--share-device", action="store_true", help="dry run: every rank uses GPU 0 (with --backend gloo)")
    ap.add_argument("--gather", default="auto", choices=["auto", "native", "torch"], help="who runs the record all-gather: native = the C-ABI's "
                    "own RCCL call (dvo_hip_gather_records_*, csrc/gather_rccl.hip: what a C++ host uses), torch = torch.distributed.all_gather "
                    "(dvo_slam_amd/parallel.py::RecordGatherer); auto = native with --backend nccl, torch otherwise (gloo dry runs)")
    ap.add_argument("--force-gather", action="store_true", help="run the N > 1 record path (process group, pinned staging, asynchronous all-gather, "
                    "drain) even with one rank: exercises the RCCL code path on a one-GPU box")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started as plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run on a free
        # local port -- and pass its exit code on.  Rank 0's JSON line is the only thing the ranks print on stdout.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    import dvo_slam_amd as d
    from dvo_slam_amd import datagen, parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and libdvo_hip has no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    gathering = world > 1 or args.force_gather
    if gathering:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                      # (a lone rank started without the launcher)
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    comm_dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    n_total = args.pairs
    my_pairs = parallel.shard_indices(n_total, rank, world)      # pair i -> rank i mod N
    B = len(my_pairs)
    if B < 1:
        raise SystemExit("fewer pairs than ranks")
    # synthetic input, identical bytes on the CPU and GPU sides (seed = global pair index)
    if world > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(16, (os.cpu_count() or 8) // world))) as ex:   # ctypes releases the GIL
            batches = list(ex.map(lambda i: datagen.synth_batch(i, 1, W, H, nthreads=1), my_pairs))
    else:
        batches = [datagen.synth_batch(0, B, W, H, nthreads=min(32, os.cpu_count() or 8))]
    pairs_np = {k: np.concatenate([b[k] for b in batches]) for k in ("grey_ref", "depth_ref", "grey_cur", "depth_cur", "xi_true")}
    pairs_np["K"] = batches[0]["K"]

    dev = torch.device("cuda", local_rank)
    grey = torch.from_numpy(np.concatenate([pairs_np["grey_ref"], pairs_np["grey_cur"]])).to(dev)                     # [2B,H,W] u8
    depth = torch.from_numpy(np.concatenate([pairs_np["depth_ref"], pairs_np["depth_cur"]]).view(np.int16)).to(dev)   # [2B,H,W] u16 bits
    torch.cuda.synchronize()
    grey_ptrs = [grey[i].data_ptr() for i in range(2 * B)]
    depth_ptrs = [depth[i].data_ptr() for i in range(2 * B)]

    ctx = d.Context(local_rank)
    if args.rows_per_wave:
        ctx.set_option("rows_per_wave", args.rows_per_wave)
    if args.resident_group:
        ctx.set_option("resident_group", args.resident_group)
    if args.resident != -1:
        ctx.set_option("resident", args.resident)
    if args.resident_rows:
        ctx.set_option("resident_rows", args.resident_rows)
    base_variant = 8                                                   # the library's default sweep schedule (dvo_hip.h, option "variant")
    for kv in args.option:
        key, _, value = kv.partition("=")
        ctx.set_option(key, int(value))
        if key == "variant":
            base_variant = int(value)                                  # ... unless this run asks for another: every leg below runs on it
    if args.iters_per_sync:
        ctx.set_option("iters_per_sync", args.iters_per_sync)
    if not args.no_overlap:
        # the next batch is built in the background of the current match
        # (one workgroup per compute unit: the library's policy for the device, counter "background_build_workgroups" -- 256 on MI355X)
        ctx.set_option("build_workgroups", ctx.counter("background_build_workgroups") if args.build_workgroups < 0 else args.build_workgroups)
    cam = d.RgbdCameraPyramid(W, H, pairs_np["K"], ctx)
    cam.build(4)
    # two frame sets: while batch k is being aligned, batch k+1 is re-ingested and its pyramids / role planes are built on the
    # context's build stream (a streaming pipeline's steady state); --no-overlap runs build and match back to back on one set
    n_sets = 1 if args.no_overlap else 2
    sets = [[cam.create_raw_device(grey_ptrs[i], depth_ptrs[i]) for i in range(2 * B)] for _ in range(n_sets)]
    frames = sets[0]
    refs, curs = frames[:B], frames[B:]
    # handle / address arrays of the streaming loop are built once (a C++ caller has them as plain arrays anyway)
    ref_sets = [d.FrameSet(fs[:B]) for fs in sets]
    cur_sets = [d.FrameSet(fs[B:]) for fs in sets]
    g_ref, z_ref = d.device_pointer_array(grey_ptrs[:B]), d.device_pointer_array(depth_ptrs[:B])
    g_cur, z_cur = d.device_pointer_array(grey_ptrs[B:]), d.device_pointer_array(depth_ptrs[B:])
    cfg_kwargs = dict(first_level=3, last_level=0, max_iterations=100, precision=5e-7, mu=0.0)
    cfg = d.Config(FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0.0)
    tracker = d.DenseTracker(cfg, ctx)
    last = {}

    counter = [0]

    # The streaming loop is one foreign call per step (dvo_slam_amd/apps/stream_pipeline.cpp, a C function on the C-ABI): ingest +
    # pyramids + the planes of the role each frame plays for the NEXT batch, from HBM-resident raw planes on the build stream
    # (asynchronous), then the alignment of the current batch (synchronous: returns when the transforms are on the host).
    from dvo_slam_amd.stream import StreamPipeline
    pipe = StreamPipeline(ctx, cfg, ref_sets, cur_sets, grey_ptrs[:B], depth_ptrs[:B], grey_ptrs[B:], depth_ptrs[B:])

    def build(k):
        pipe.step(now=None, nxt=k)

    def step():
        k = counter[0] % n_sets
        counter[0] += 1
        res = pipe.step(now=k, nxt=(k + 1) % n_sets)                    # next batch (with one set: this batch, built right before its match)
        last["T"] = res["transformation"].reshape(B, 4, 4)
        last["information"] = res["information"].reshape(B, 6, 6)
        last["loglik"] = res["loglik"]
        if gathering:
            # the records of this batch travel (one all-gather, RCCL) while the next batch is aligned; buffers allocated once
            rec = pipe.records()                                        # packed by the pipeline object (stream_pipeline.cpp)
            if pending[0] is not None:
                gathered[0] = pending[0].result()
            pending[0] = gatherer.start(rec)
        return None

    pending, gathered = [None], [None]
    gather_kind = ("native" if args.backend == "nccl" else "torch") if args.gather == "auto" else args.gather
    gatherer = None
    if gathering and gather_kind == "native":
        # the communicator's 128-byte id travels once through the launcher's process group; every gather after that is the library's own
        # A rank on which the library cannot set the communicator up (no librccl beside the library, an id that did not arrive) must not
        # leave the others waiting inside a collective: every step below is followed by an agreement of all ranks, and without it all of
        # them take torch.distributed's all-gather instead (reported in the line: config.parallelism).
        def agreed(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=comm_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())
        uid, why = torch.zeros(128, dtype=torch.uint8, device=comm_dev), ""
        if rank == 0:
            try:
                uid = torch.frombuffer(bytearray(parallel.NativeRecordGatherer.unique_id(ctx)), dtype=torch.uint8).to(comm_dev)
            except Exception as e:                                      # noqa: BLE001 -- whatever it is, the ranks must agree on it
                why = str(e)
        dist.broadcast(uid, src=0)
        if agreed(bool(uid.any().item())):
            try:
                gatherer = parallel.NativeRecordGatherer(ctx, bytes(uid.cpu().numpy().tobytes()), n_total, rank, world)
            except Exception as e:                                      # noqa: BLE001
                why = str(e)
            if not agreed(gatherer is not None):
                if gatherer is not None:
                    gatherer.close()
                gatherer = None
        if gatherer is None:
            if why:
                print("bench.py: rank %d: native record gather unavailable (%s); all ranks use torch.distributed" % (rank, why), file=sys.stderr)
            gather_kind = "torch"
            gatherer = parallel.RecordGatherer(n_total, rank, world, device=comm_dev)
    elif gathering:
        gatherer = parallel.RecordGatherer(n_total, rank, world, device=comm_dev)

    def drain():
        if pending[0] is not None:
            gathered[0] = pending[0].result()
            pending[0] = None

    def barrier():
        torch.cuda.synchronize()
        if gathering:
            dist.barrier()

    # Lanes (round 6): with two pairs per compute unit and more the rank's pairs are dealt to 2-3 lanes -- a context and a host thread
    # each, running their shards' steps without waiting for one another (dvo_slam_amd/apps/stream_pipeline.cpp, dvo_stream_lanes_*): one
    # lane's latency-bound phases pass beside another's sweeps.  A step is still one full ingest + alignment of EVERY pair of the rank;
    # its results are collected in submission order, at most LANE_DEPTH steps behind.
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    # (measured on 256 compute units, alternated on one box: 1024 pairs 11.3 -> 10.8 ms per step with two lanes, 10.9-11.2 with three, 11.1 with
    # four -- while the host thread still spent 0.8 ms per step in front of the first launch, which a second lane hid; with that fixed
    # (capi_frames.inc::ensure_roles) two lanes are within +-2 % of one, box by box: 10.9-11.0 against 11.05-11.3 on one, 11.2 against 11.07 on
    # another.  The timed loop therefore runs on ONE lane unless --lanes asks for more; from three and a half pairs per compute unit on the
    # two-lane loop is timed as a leg of its own and reported under "lanes")
    lanes_leg = args.lanes == 0 and not args.no_overlap and not args.no_lanes_leg and not args.loop_only and 2 * B >= 7 * cus and world == 1
    n_lanes = args.lanes if args.lanes > 0 else (2 if lanes_leg else 1)
    n_lanes = max(1, min(n_lanes, B // 64 if B >= 128 else 1))
    LANE_DEPTH = max(1, args.lane_depth)
    lanes = None
    if n_lanes > 1:
        from dvo_slam_amd.stream import StreamLanes
        # (the lanes' background builds share the chip: 3/4 workgroup per compute unit each for two lanes -- 192 on MI355X -- 3/8 for three)
        per_cu = ctx.counter("background_build_workgroups")
        lane_cap = (per_cu * 3 // 4 if n_lanes == 2 else max(64, per_cu * 3 // 2 // n_lanes)) if args.build_workgroups < 0 else args.build_workgroups
        lane_ctx, lane_cam = [], {}
        for _ in range(n_lanes):
            c = d.Context(local_rank)
            for kv in args.option:
                key, _, value = kv.partition("=")
                c.set_option(key, int(value))
            c.set_option("build_workgroups", lane_cap)
            lane_ctx.append(c)
            lane_cam[id(c)] = d.RgbdCameraPyramid(W, H, pairs_np["K"], c)
            lane_cam[id(c)].build(4)

        def make_frames(c, idx):
            cam_l = lane_cam[id(c)]
            return ([cam_l.create_raw_device(grey_ptrs[i], depth_ptrs[i]) for i in idx],
                    [cam_l.create_raw_device(grey_ptrs[B + i], depth_ptrs[B + i]) for i in idx])
        lanes = StreamLanes(lane_ctx, cfg, B, make_frames, grey_ptrs[:B], depth_ptrs[:B], grey_ptrs[B:], depth_ptrs[B:], depth=LANE_DEPTH)

    def lanes_collect():
        res = lanes.collect()
        last["T"] = res["transformation"].reshape(B, 4, 4).copy()
        last["information"] = res["information"].reshape(B, 6, 6).copy()
        last["loglik"] = res["loglik"].copy()
        if gathering:
            rec = lanes.records()
            if pending[0] is not None:
                gathered[0] = pending[0].result()
            pending[0] = gatherer.start(rec)

    def lanes_loop(k_steps):
        for _ in range(k_steps):
            lanes.submit()
            if lanes.outstanding >= LANE_DEPTH:
                lanes_collect()
        while lanes.outstanding:
            lanes_collect()

    if n_sets > 1:
        build(0)                                                        # prime the pipeline: step k aligns set k % 2 and builds the other
    elapsed_one_lane = elapsed_lanes = lanes_vs_one = None

    def one_lane_loop():
        for _ in range(args.warmup):
            step()
        drain()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()                                                         # the last batch's records have arrived on every rank
        barrier()
        return time.perf_counter() - t0

    def lanes_timed_loop():
        lanes_loop(max(2, args.warmup))
        drain()
        barrier()
        t0 = time.perf_counter()
        lanes_loop(args.steps)                                          # EXACTLY args.steps steps: submitted, aligned and collected inside the timed region
        drain()
        barrier()
        return time.perf_counter() - t0

    if lanes is not None and not lanes_leg:                             # --lanes N: the timed loop runs on the lanes
        elapsed_one_lane = one_lane_loop()
        one_lane_T = last["T"].copy()
        elapsed = elapsed_lanes = lanes_timed_loop()
        lanes_vs_one = float(np.abs(parallel.twists_of(last["T"]) - parallel.twists_of(one_lane_T)).max())
    else:
        elapsed = elapsed_one_lane = one_lane_loop()
        if lanes is not None:                                           # the two-lane loop as a leg of its own (outside `value`)
            one_lane_T = last["T"].copy()
            keep_last = dict(last)
            elapsed_lanes = lanes_timed_loop()
            lanes_vs_one = float(np.abs(parallel.twists_of(last["T"]) - parallel.twists_of(one_lane_T)).max())
            last.update(keep_last)
    if lanes is not None:
        lanes.close()                                                   # (its frames and contexts go with it: the legs below run on the rank's own context)
        lanes = None
        del lane_ctx, lane_cam
    if gathering:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- everything below is outside the timed region -------------------------------------------------------
    if args.loop_only:
        if gathering:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "frame-pair alignments/s (640x480, 4-level pyramid) + finest-level HBM GB/s vs roofline",
                              "value": round(n_total * args.steps / elapsed, 2), "unit": "alignments/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "loop_only": True,
                              "config": {"pairs": n_total, "pairs_per_gpu": B, "lanes": n_lanes},
                              "counters": {k: ctx.counter(k) for k in ("overlapped_tails", "overlapped_steps", "tail_drains", "tail_wait_us")},
                              "ms_per_step_one_lane": None if elapsed_lanes is None else round(elapsed_one_lane / args.steps * 1e3, 3)}))
        return
    twist_err = float(np.abs(parallel.twists_of(last["T"]) - pairs_np["xi_true"]).max())
    if args.records_out and rank == 0:
        full = gathered[0] if gathering else parallel.pack_records(parallel.twists_of(last["T"]), last["information"], last["loglik"])
        np.save(args.records_out, full)
    nan_results = int((~np.isfinite(last["T"]).all(axis=(1, 2)) | ~np.isfinite(last["information"]).all(axis=(1, 2))).sum())
    t_match0 = time.perf_counter()
    tracker.match_batch_arrays(refs, curs)
    match_only_ms = (time.perf_counter() - t_match0) * 1e3

    # roofline of the dominant kernel: finest-level fused warp/residual/Jacobian/reduce sweep, HIP events on the context stream
    # (timed where the sweeps of a match run: after three Gauss-Newton steps on the level, i.e. at the converged transform with the
    # t-distribution weights on -- not at the identity with unit weights)
    # (round 4: the MEAN of 24 back-to-back launches per level -- the closest a run without the profiler gets to the in-situ average of
    # profiles/r04_bench_kernel_stats_insitu.txt; rounds 1-3 reported the smaller of two means of 10)
    ROOFLINE_REPS = 24
    k_ms = {lvl: tracker.time_residual_kernel(refs, curs, lvl, reps=ROOFLINE_REPS, warm_iterations=3) for lvl in (0, 1, 2, 3)}
    # the same launch on the other schedules of the finest level (library option "variant"): 7 = the window sweep whose residuals equal the
    # oracle's bit for bit (no contraction, correctly rounded divisions), 6 = 7 with the f32 Gram (no f16 operands anywhere)
    k_ms_variants = {}
    for v in (7, 6):
        ctx.set_option("variant", v)
        k_ms_variants[v] = tracker.time_residual_kernel(refs, curs, 0, reps=ROOFLINE_REPS, warm_iterations=3)
    ctx.set_option("variant", base_variant)                           # (what the run was started with, not a constant)
    ctx.set_option("gram_lo_parts", 0)                                 # ... and round 5's default: the Jacobian components as f16 high parts alone
    k_ms_hi_j = tracker.time_residual_kernel(refs, curs, 0, reps=ROOFLINE_REPS, warm_iterations=3)
    ctx.set_option("gram_lo_parts", 1)
    stream_ms = tracker.time_stream_mix(refs, curs, 0, reps=10)                       # the same planes streamed in pixel order, read only
    stream_w_ms = tracker.time_stream_mix(refs, curs, 0, reps=10, with_write=True)   # ... plus the 8-B residual pair the sweep writes
    algo_bytes = ALGO_BYTES_PER_PIXEL * W * H * B
    achieved = algo_bytes / (k_ms[0] * 1e-3) / 1e9
    traffic = _pmc_traffic(B)
    roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBPS, 4),
                    traffic=traffic, kernel=_kernel_label(base_variant, B),
                    kernel_ms=round(k_ms[0], 4), kernel_ms_is="mean of %d back-to-back launches (HIP events on the context stream)" % ROOFLINE_REPS,
                    kernel_ms_hi_j=round(k_ms_hi_j, 4), frac_hi_j=round(algo_bytes / (k_ms_hi_j * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    kernel_ms_exact_arithmetic=round(k_ms_variants[7], 4), kernel_ms_f32_gram=round(k_ms_variants[6], 4),
                    frac_exact_arithmetic=round(algo_bytes / (k_ms_variants[7] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    frac_f32_gram=round(algo_bytes / (k_ms_variants[6] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    schedules_note="kernel_ms / frac: the shipped schedule (variant 8: fused multiply-adds, v_rcp_f32 in the projection; residuals within "
                                   "2e-5 of the oracle's, tests/test_gpu_parity.py::test_contracted_sweep_against_the_exact_one; every Gram operand an "
                                   "exact f16 high + low pair, f32 accumulation: 22-bit operands, the precision class of the reference's f32 sums); "
                                   "kernel_ms_hi_j / frac_hi_j: option gram_lo_parts 0, round 5's default -- the twelve Jacobian components as f16 high "
                                   "parts alone (11-bit operands) at this level: faster, and NOT the number to credit; "
                                   "kernel_ms_exact_arithmetic: variant 7, residuals and constraint counts bit-identical to the oracle's MATH mode, "
                                   "f16 hi + lo Gram operands; kernel_ms_f32_gram: variant 6, the same with the f32 matrix instruction (no f16 anywhere)",
                    algorithmic_bytes_per_launch=algo_bytes,
                    algorithmic_bytes_per_pixel=ALGO_BYTES_PER_PIXEL,
                    moved_bytes_per_pixel=None if traffic is None else round(traffic / (W * H * B), 2),
                    moved_GBps=None if traffic is None else round(traffic / (k_ms[0] * 1e-3) / 1e9, 1),
                    moved_frac=None if traffic is None else round(traffic / (k_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    limited_by="vector-instruction issue (profiles/r06_pmc_finest_sweep.txt: 183 vector instructions per 64-pixel row with every low part kept, the vector "
                               "ALU active ~80 % of the kernel's cycles, the LDS array ~68 %); `achieved` / `frac` price the launch at SURVEY.md 8(d)'s 40 ALGORITHMIC bytes per pixel -- a "
                               "derived rate, the contract's definition -- while the bytes the kernel really moves (`traffic`, PMC) give `moved_GBps` / "
                               "`moved_frac`: the kernel is not near any memory limit",
                    timed_at="converged transform, t-distribution weights on (3 warm-up Gauss-Newton steps on the level)",
                    bare_stream_ms=round(stream_ms, 4), bare_stream_frac=round(algo_bytes / (stream_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    bare_stream_with_write_ms=round(stream_w_ms, 4),
                    bare_stream_note="kernels that only read the planes the sweep reads (reference {Zsel, I} 8 B + current {I, Z} 8 B per pixel) in "
                                     "pixel order (no window, arithmetic or reduction; the second also writes the 8-B residual pair), timed the same "
                                     "way: what this part's memory system needs for the bytes the sweep moves; bare_stream_frac is that time "
                                     "priced at the 40 algorithmic bytes",
                    per_level_kernel_ms={str(k): round(v, 4) for k, v in k_ms.items()},
                    per_kernel=_per_kernel_rooflines(k_ms, B))

    # latency of BASELINE config 2: one 640x480 pair, 4 levels
    one = d.Result()
    lat = []
    for _ in range(12):
        t1 = time.perf_counter()
        tracker.match(refs[0], curs[0], one, with_stats=False)
        lat.append((time.perf_counter() - t1) * 1e3)
    single_pair_ms = float(np.median(lat[2:]))

    # ... of the two pairs the tracking front end aligns per frame (dvo_slam/src/local_tracker.cpp:180-184), in this configuration
    # and in the front end's own (dvo_ros/cfg/dvo.cfg defaults = dvo_benchmark/launch/benchmark.yaml: levels 3..1, Precision 1e-4, Mu 0.05,
    # initial estimate), and of one pair on the launch-per-step path (option resident 0) for comparison
    def median_ms(fn, reps=12):
        t = []
        for _ in range(reps):
            t1 = time.perf_counter()
            fn()
            t.append((time.perf_counter() - t1) * 1e3)
        return round(float(np.median(t[2:])), 3)
    latency = {"pairs_1": round(single_pair_ms, 3), "pairs_2": median_ms(lambda: tracker.match_batch_arrays(refs[:2], curs[:2]))}
    front_end = d.DenseTracker(d.Config(FirstLevel=3, LastLevel=1, MaxIterationsPerLevel=50, Precision=1e-4, Mu=0.05, UseInitialEstimate=True), ctx)
    guess = np.stack([np.eye(4)] * len(refs[:2]))
    latency["front_end_config_pairs_2"] = median_ms(lambda: front_end.match_batch_arrays(refs[:2], curs[:2], T_init=guess))
    ctx.set_option("resident", 0)
    latency["pairs_1_launch_path"] = median_ms(lambda: tracker.match(refs[0], curs[0], one, with_stats=False))
    ctx.set_option("resident", -1)
    # ... and in the reference-compatible mode (option ref_compat), which the resident kernel carries since round 4
    ctx.set_option("ref_compat", 1)
    latency["ref_compat_pairs_1"] = median_ms(lambda: tracker.match(refs[0], curs[0], one, with_stats=False))
    latency["ref_compat_front_end_config_pairs_2"] = median_ms(lambda: front_end.match_batch_arrays(refs[:2], curs[:2], T_init=guess))
    ctx.set_option("ref_compat", 0)

    # What the one-GPU measurements predict for N GPUs (strong scaling of the fixed 1024-pair total: every rank gets pairs / N): the
    # same streaming loop timed on the first pairs / N pairs of this rank.  The driver measures the real curve; this is the prediction
    # it can be checked against (it leaves out the record all-gather, which travels under the next step).
    scaling_model = None
    if world == 1 and n_sets > 1 and not args.no_scaling_model:
        per_gpu = {}
        for n_gpus in (1, 2, 4, 8):
            b = B // n_gpus
            if b < 1:
                continue
            if n_gpus == 1:
                per_gpu[n_gpus] = elapsed / args.steps * 1e3
                continue
            sub = StreamPipeline(ctx, cfg, [d.FrameSet(fs[:b]) for fs in sets], [d.FrameSet(fs[B:B + b]) for fs in sets],
                                 grey_ptrs[:b], depth_ptrs[:b], grey_ptrs[B:B + b], depth_ptrs[B:B + b])
            sub.step(now=None, nxt=0)
            for j in range(2):
                sub.step(now=j % 2, nxt=(j + 1) % 2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = max(args.steps, 10)
            for j in range(reps):
                sub.step(now=j % 2, nxt=(j + 1) % 2)
            torch.cuda.synchronize()
            per_gpu[n_gpus] = (time.perf_counter() - t1) / reps * 1e3
        scaling_model = {"ms_per_step_at_pairs_per_gpu": {str(B // n): round(ms, 3) for n, ms in per_gpu.items()},
                         "predicted_alignments_per_s": {str(n): round(n_total / (ms * 1e-3), 1) for n, ms in per_gpu.items()},
                         "predicted_efficiency": {str(n): round(per_gpu[1] / (n * ms), 3) for n, ms in per_gpu.items()},
                         "note": "one GPU, HBM-resident loop on pairs/N pairs; N ranks run this independently (no data-path collective), so "
                                 "alignments/s at N GPUs ~ total pairs / that step time"}
        pipe.step(now=None, nxt=counter[0] % n_sets)      # (the sub-pipelines re-ingested some frames of the sets: restore the pipeline's state)

    # The reference-compatible mode (option "ref_compat": projection and weights multiply with the host CPU's _mm_rcp_ps like the reference's
    # SSE path) on the same streaming loop: the same schedule, two table lookups per pixel
    ref_compat = None
    if world == 1 and not args.no_ref_compat:
        ctx.set_option("ref_compat", 1)
        pipe.step(now=None, nxt=counter[0] % n_sets)
        for j in range(2):
            step()
        barrier()
        t1 = time.perf_counter()
        for j in range(args.steps):
            step()
        barrier()
        el_c = time.perf_counter() - t1
        ref_compat = {"value": round(n_total * args.steps / el_c, 2), "unit": "alignments/s", "ms_per_step": round(el_c / args.steps * 1e3, 3),
                      "max_twist_error_vs_truth": float(np.abs(parallel.twists_of(last["T"]) - pairs_np["xi_true"]).max()),
                      "note": "the same HBM-resident loop with option ref_compat on (the opt-in mode whose trajectories follow the reference's own, "
                              "DESIGN.md section 2); the finest-level sweep is then dvo_hip::k_sweep_fast<2, false, true, 2, false, 0>: the contracted window sweep with the "
                              "host CPU's _mm_rcp_ps table in projection and weights (round 5; rounds 3-4: k_sweep_window<true, true, 4>, "
                              "still the bit-exact anchor of the mode under option variant 7)"}
        ctx.set_option("ref_compat", 0)
        pipe.step(now=None, nxt=counter[0] % n_sets)

    # The f16 range guard under load (round 5): 1 % of the pairs carry a depth step that no f16 Gram operand represents (2 mm / 10 m
    # checkerboard, both frames) -- those pairs, and only those, are repeated with the f32 Gram as a batch of their own
    guard_stress = None
    if world == 1 and not args.no_guard_stress and B >= 8:
        k_bad = max(1, B // 100)
        bad = [int(round((j + 0.5) * B / k_bad)) for j in range(k_bad)]
        yy, xx = np.mgrid[0:H, 0:W]
        step_depth = np.where(((xx // 8) + (yy // 8)) % 2 == 0, 10, 50000).astype(np.uint16).view(np.int16)   # raw counts at 1/5000 m
        step_dev = torch.from_numpy(step_depth).to(dev)
        keep = [(i, depth[i].clone(), depth[B + i].clone(), grey[B + i].clone()) for i in bad]
        for i in bad:
            depth[i].copy_(step_dev)
            depth[B + i].copy_(step_dev)
            grey[B + i].copy_(grey[i])
        torch.cuda.synchronize()
        pipe.step(now=None, nxt=counter[0] % n_sets)
        for j in range(2):
            step()
        barrier()
        r0 = ctx.counter("f16_range_repeats")
        t1 = time.perf_counter()
        for j in range(args.steps):
            step()
        barrier()
        el_g = time.perf_counter() - t1
        guard_stress = {"pairs_out_of_range": k_bad, "ms_per_step": round(el_g / args.steps * 1e3, 3),
                        "ratio_to_clean_step": round(el_g / elapsed_one_lane, 4),
                        "pairs_repeated_per_step": round((ctx.counter("f16_range_repeats") - r0) / args.steps, 2),
                        "note": "the timed loop with %d of the %d pairs replaced by a 2 mm / 10 m depth checkerboard (Jacobian components beyond +-65504): "
                                "the sweep raises the pair's word, the library repeats those pairs alone with the f32 Gram" % (k_bad, B)}
        for i, dr, dc, gc in keep:
            depth[i].copy_(dr)
            depth[B + i].copy_(dc)
            grey[B + i].copy_(gc)
        torch.cuda.synchronize()
        pipe.step(now=None, nxt=counter[0] % n_sets)
        step()                                                          # (both frame sets hold the bench's own pairs again)
        step()
        barrier()

    # PCIe-inclusive leg (never `value`): the same pipeline, but every step's raw planes are handed over in pinned HOST memory
    # (SURVEY.md 8d config 4 "incl. H2D of 2 planes per frame") -- DMA on the upload stream, build on the build stream, match on
    # the main stream, three batches in flight
    from_host = None
    if not args.no_from_host:
        pinned = d.PinnedRawPlanes(ctx, 2 * B, W, H)
        for i in range(B):
            pinned.grey[i][:] = pairs_np["grey_ref"][i]
            pinned.grey[B + i][:] = pairs_np["grey_cur"][i]
            pinned.depth[i][:] = pairs_np["depth_ref"][i]
            pinned.depth[B + i][:] = pairs_np["depth_cur"][i]

        # the same C loop as the timed one, fed from the host arrays (one foreign call per step: with 1024 pairs per step building
        # 4096 ctypes entries per step in Python had become a quarter of the step)
        pipe.set_host_planes(pinned.grey[:B], pinned.depth[:B], pinned.grey[B:], pinned.depth[B:])

        def host_build(k):
            pipe.step_host(now=None, nxt=k)

        def host_step(j):
            res = pipe.step_host(now=j % n_sets, nxt=(j + 1) % n_sets)
            return {"T": res["transformation"].reshape(B, 4, 4)}

        if n_sets > 1:
            host_build(0)
        for j in range(2):
            host_step(j)
        barrier()
        keys = ("host_batches", "host_ns_prepare", "host_ns_enqueue", "host_ns_wait", "host_ns_finish")
        c0 = [ctx.counter(k) for k in keys]
        t_h = time.perf_counter()
        for j in range(args.steps):
            out_h = host_step(j)
        barrier()
        el_h = time.perf_counter() - t_h
        c1 = [ctx.counter(k) for k in keys]
        host_split = {k[8:]: round((b - a) / max(c1[0] - c0[0], 1) / 1e6, 3) for k, a, b in zip(keys[1:], c0[1:], c1[1:])}
        if world > 1:
            t = torch.tensor([el_h], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_h = float(t.item())
        bytes_step = 2 * B * W * H * 3
        from_host = {"value": round(n_total * args.steps / el_h, 2), "unit": "alignments/s", "ms_per_step": round(el_h / args.steps * 1e3, 3),
                     "h2d_bytes_per_step_per_gpu": bytes_step, "h2d_GBps_per_gpu": round(bytes_step * args.steps / el_h / 1e9, 2),
                     "same_results": bool(np.array_equal(out_h["T"], last["T"])),
                     "roofline": {"bound": "pcie", "achieved": round(bytes_step * args.steps / el_h / 1e9, 2), "peak": PCIE_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(bytes_step * args.steps / el_h / 1e9 / PCIE_PEAK_GBPS, 4),
                                  "note": "host-to-device bytes of the raw planes per second per GPU against the PCIe Gen5 x16 spec rate: this leg "
                                          "is bound by the link, not by a kernel"},
                     "host_thread_ms_per_match_call": host_split,
                     "note": "raw planes (u8 grey + u16 depth of both frames of every pair, 1.84 MB per pair) DMA-ed from pinned host "
                             "memory every step; reported beside `value`, never as it"}
        d.upload_wait(ctx)
        pinned.close()

    out = None
    if rank == 0:
        value = n_total * args.steps / elapsed
        out = {
            "metric": "frame-pair alignments/s (640x480, 4-level pyramid) + finest-level HBM GB/s vs roofline",
            "value": round(value, 2), "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (Gram operands exact f16 hi + lo pairs on every level, f32 accumulate)", "data": "synthetic",
            "config": {"workload": "%d pairs (BASELINE config 4): independent 640x480 RGB-D frame pairs, seeds 0..%d, pair i on rank i mod %d, "
                                   "4-level pyramid, FirstLevel=3, LastLevel=0, MaxIterationsPerLevel=100, Precision=5e-7, Mu=0; "
                                   "step = every rank re-ingests the raw planes of its shard from HBM (pyramids, sampling planes, point "
                                   "selection) and aligns it as one batch%s" % (
                                       n_total, n_total - 1, world, "; build and match run back to back" if args.no_overlap else
                                       "; the build of step k+1 (build stream) overlaps the match of step k, every step does one full build and one full match"),
                       "pairs": n_total, "pairs_per_gpu": B, "width": W, "height": H,
                       "lanes": ("%d lanes per GPU: the rank's pairs dealt to %d contexts of its GPU, a host thread each, that run their shards' steps out of phase "
                                 "(dvo_stream_lanes_*, dvo_slam_amd/apps/stream_pipeline.cpp); a step = one full ingest + alignment of every pair, collected in "
                                 "submission order at most %d steps behind" % (n_lanes, n_lanes, LANE_DEPTH)) if n_lanes > 1 and not lanes_leg else "1 lane",
                       "parallelism": "independent pairs sharded round-robin over %d GPU(s) (fixed total: strong scaling), one all-gather of "
                                      "256-B records per step%s" % (world, "" if not gathering else
                                                                    " (ncclAllGather called by the C-ABI: dvo_hip_gather_records_*)" if gather_kind == "native"
                                                                    else " (torch.distributed.all_gather)"),
                       # (what the first keys of this line do NOT say by themselves -- SURVEY.md 8d quotes config 4 "incl. H2D": that is contract_value)
                       "ingest": "hbm-resident raw planes (u8 grey + u16 depth per frame already in HBM when the timed region starts; the same loop "
                                 "fed from pinned host memory, PCIe-inclusive, is contract_value)"},
            "roofline": roofline,
            "lanes": None if elapsed_lanes is None else {
                "lanes": n_lanes, "ms_per_step": round(elapsed_lanes / args.steps * 1e3, 3), "value": round(n_total * args.steps / elapsed_lanes, 2),
                "ms_per_step_one_lane": round(elapsed_one_lane / args.steps * 1e3, 3), "timed_loop_runs_on": "one lane" if lanes_leg else "%d lanes" % n_lanes,
                "build_workgroups_per_lane": lane_cap, "max_twist_difference_to_one_lane": lanes_vs_one,
                "note": "the streaming loop over several contexts of the GPU, a host thread each, that run their shards' steps out of phase "
                        "(dvo_stream_lanes_*, dvo_slam_amd/apps/stream_pipeline.cpp), timed like the main loop (EXACTLY `steps` full steps, submitted and "
                        "collected inside the timed region).  Within +-2 % of the one-lane loop box by box since the host thread no longer spends "
                        "0.8 ms per step in front of the first launch (round 6): reported, not `value`, unless --lanes asks for it"},
            "match_only_ms_per_batch": round(match_only_ms, 3),
            "single_pair_ms": round(single_pair_ms, 3),
            "latency_ms": latency,
            "contract_value": None if from_host is None else from_host["value"],
            "contract_value_note": "SURVEY.md 8d config 4 / BASELINE.md: alignments/s INCLUDING the host-to-device transfer of two raw planes per "
                                   "frame (= from_host.value, PCIe-bound, see from_host.roofline); `value` has the raw planes resident in HBM",
            "from_host": from_host,
            "ref_compat": ref_compat,
            "f16_guard_stress": guard_stress,
            "scaling_model": scaling_model,
            "config3": ("unmeasured: DVO_TUM_ROOT unset (no TUM RGB-D sequence on this box; tests/test_gpu_replay.py -k real replays one when it is)"
                        if not os.environ.get("DVO_TUM_ROOT") else "DVO_TUM_ROOT=%s: see tests/test_gpu_replay.py -k real" % os.environ["DVO_TUM_ROOT"]),
            "max_twist_error_vs_truth": twist_err, "nan_results": nan_results,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pairs_np, cfg_kwargs, args.cpu_sample_pairs, args.cpu_matches_per_thread, args.cpu_trials)
    if gathering:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if out is not None and args.force_gather:
            out["forced_gather"] = {"backend": args.backend, "world_size": world, "gather": gather_kind}
        print(json.dumps(out))


def _kernel_label(variant, pairs):
    """Name of the finest-level sweep kernel the schedule `variant` launches (launch_residual_reduce, align_common.h), as rocprofv3 prints it."""
    names = {8: "dvo_hip::k_sweep_fast<2, false, true, 0, false, 0> (template arguments: operand stores without lane swaps, no partial tile column, "
                "packed residual pairs, not the ref_compat arithmetic, every operand an f16 high + low pair, no solver step in the tail)", 9: "dvo_hip::k_sweep_fast<1, false, true, 0, false, 0>", 7: "dvo_hip::k_sweep_window<true, false, 4>",
             6: "dvo_hip::k_sweep_window<false, false, 4>", 5: "dvo_hip::k_residual_reduce_mfma", 0: "dvo_hip::k_residual_reduce"}
    what = ("64 x 16 tiles, the current frame's {I, Z} window staged in LDS, contracted f32 pixel arithmetic, Gram accumulation on the f16 matrix "
            "pipe: every component as an exact hi + lo pair") if variant >= 8 else "option variant=%d, see include/dvo_hip.h" % variant
    return "%s (pyramid level 0, %d pairs per launch; %s)" % (names.get(variant, "variant %d" % variant), pairs, what)


def _per_kernel_rooflines(k_ms, pairs):
    """Every kernel that takes 3 % of the step or more, with its roofline: `achieved` / `frac` price a full launch at its ALGORITHMIC bytes
    (the 40 B per level-pixel of SURVEY.md 8(d) for the sweeps) -- for the four sweep levels from the launch times
    measured live above (HIP events), for the others from the in-situ rocprofv3 trace; `moved_GBps` / `moved_frac` are the bytes the
    kernel really moved (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes over the bench loop: scripts/r6_rooflines.sh ->
    profiles/r06_kernel_rooflines.json, which bench.py cannot collect itself)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r06_kernel_rooflines.json")))
    except (OSError, ValueError):
        return None
    if rec.get("pairs") != pairs:
        return None
    level_of = {300 * pairs: 0, 75 * pairs: 1, 24 * pairs: 2}
    out = []
    for k in rec["kernels"]:
        if k["step_share"] < 0.03:
            continue
        e = {key: k.get(key) for key in ("kernel", "workgroups", "ms", "ms_alone", "step_share", "achieved_GBps", "frac", "moved_GBps", "moved_frac",
                                         "valu_active_share", "limited_by")}
        e["ms_from"] = "rocprofv3 kernel trace of the bench loop (in situ)"
        lvl = level_of.get(k["workgroups"]) if k["kernel"].startswith("k_sweep_fast") else (3 if k["kernel"].startswith("k_residual_reduce_mfma") else None)
        if lvl is not None and k.get("algorithmic_bytes"):
            e["ms_live"] = round(k_ms[lvl], 4)
            e["achieved_GBps"] = round(k["algorithmic_bytes"] / (k_ms[lvl] * 1e-3) / 1e9, 1)
            e["frac"] = round(e["achieved_GBps"] / HBM_PEAK_GBPS, 4)
            e["ms_from"] = "this run (HIP events, mean of back-to-back launches); moved_* from the committed PMC passes"
        out.append(e)
    worst = min((e for e in out if e["frac"] is not None), key=lambda e: e["frac"], default=None)
    return {"kernels": out, "worst_at_algorithmic_bytes": None if worst is None else "%s (%d workgroups): %.3f" % (worst["kernel"], worst["workgroups"], worst["frac"]),
            "source": "profiles/r06_kernel_rooflines.json + profiles/r06_kernel_rooflines.md"}


def _pmc_traffic(pairs):
    """HBM bytes per launch of the finest-level kernel from the committed rocprofv3 --pmc passes (profiles/pmc_finest_kernel.json:
    FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, separate passes, see scripts/pmc.sh).  bench.py cannot run the profiler itself.  The
    number is reported only when it was collected for the same number of pairs per launch AND the record names the very source of
    the sweep kernel that is compiled now (sha256 of align_fast.hip + gram_f16.h + sweep_parts.h + pixel_math.h) -- a stale record reads as null."""
    import hashlib
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_finest_kernel.json")))
        h = hashlib.sha256()
        for f in ("align_fast.hip", "gram_f16.h", "sweep_parts.h", "pixel_math.h"):
            h.update(open(os.path.join(ROOT, "dvo_slam_amd", "csrc", f), "rb").read())
        if rec.get("kernel_source_sha256") != h.hexdigest() or rec["pairs_per_launch"] != pairs:
            return None
        return rec["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


if __name__ == "__main__":
    main()
