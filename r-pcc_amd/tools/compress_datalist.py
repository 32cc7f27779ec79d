#!/usr/bin/env python3
"""Datalist compression -- counterpart of the reference's tools/compress_datalist.py.  Same flags
(+ --batch); frames go through the device in batches (one fused call per batch), a thread pool does
the host-side part (file reads, entropy coding, file writes) like the reference's ThreadPoolExecutor.
Output path rule of the reference (tools/compress_datalist.py:136-142): output_dir + original path with
the extension replaced by `rpcc`.

Multi-GPU: launch one process per GPU (torchrun); rank r takes datalist entries r, r+R, ... (frames are
independent; no collective on the data path).  By default every rank writes the files of its shard.  --gather is the
north star's "RCCL only for the final gather of compressed bitstreams": every rank entropy-codes its shard, the .rpcc byte
strings (~61 KB per frame) go to rank 0 in rounds (sharding.RoundGather) and rank 0 writes all files in datalist order."""
import os
import sys
import time
from concurrent import futures

BASE_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, BASE_DIR)

import numpy as np  # noqa: E402

import rpcc_amd  # noqa: E402,F401
from rpcc_amd.dataset import build_dataset  # noqa: E402
from rpcc_amd.loader import StreamingCompressor  # noqa: E402
from rpcc_amd.pipeline import BatchCompressor  # noqa: E402
from rpcc_amd.sharding import RoundGather, shard_indices  # noqa: E402
from rpcc_amd.tools.compress import apply_fps_mode, make_parser, resolve_cfg  # noqa: E402
from rpcc_amd.utils import available_cpus, frame_identity, local_rank_env, pin_rank_cpus  # noqa: E402


def output_path_for(output_dir, file_name):
    """tools/compress_datalist.py:136-142."""
    file_name = file_name.strip()
    if file_name[0] == "/":
        file_name = file_name[1:]
    out = os.path.join(output_dir, file_name)
    return out.replace(out.split(".")[-1], "rpcc")


def _prefetch(gen, depth=2):
    """Runs a generator on a background thread, `depth` items ahead (file reads overlap the device and the entropy coder)."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    END = object()

    def work():
        try:
            for item in gen:
                q.put(item)
            q.put(END)
        except BaseException as e:  # noqa: BLE001
            q.put(e)
    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is END:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


def write_blob(output_dir, name, blob):
    out = output_path_for(output_dir, name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "wb") as f:
        f.write(blob)
    return len(blob)


def init_gather_group(rank, world, local):
    """Process group of the --gather exchange: RCCL (backend "nccl") between the ranks' GPUs; RPCC_DIST_BACKEND overrides."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    backend = os.environ.get("RPCC_DIST_BACKEND", "nccl")
    kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def compress(args, streaming_factory=None):
    """streaming_factory (tests): builds the object that turns batches into .rpcc strings -- loader.StreamingCompressor's constructor
    arguments and run() -- so that the sharding / gathering / file-writing logic of this driver can run on hosts without a GPU."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = "cuda:%d" % local
    # one process per GPU: every rank keeps its own slice of the host's CPUs (the feed is host-bound: DESIGN.md section 7)
    # (only when the launcher names both LOCAL_RANK and LOCAL_WORLD_SIZE, as torchrun does: with WORLD_SIZE as the fall-back a multi-node job
    # would cut slices by the global world, and every rank of a host that lacks LOCAL_RANK would take the first one)
    lenv = local_rank_env()
    pinned = pin_rank_cpus(lenv[0], lenv[1]) if lenv is not None else None
    if pinned is not None:
        args.workers = max(1, min(args.workers, available_cpus()))
    apply_fps_mode(args)
    cfg, accuracy, segment_cfg, model_cfg, basic_compressor, uniform = resolve_cfg(args)
    dataset = build_dataset(datalist=args.datalist, lidar_type=args.lidar, device=device)
    bc = BatchCompressor(dataset.PCTransformer, cluster_num=segment_cfg["cluster_num"], accuracy=accuracy / 2,
                         ground_threshold=segment_cfg["ground_vertical_threshold"], uniform=uniform,
                         model_method=model_cfg["model_method"], compressor_cfg=dict(cfg),
                         basic_compressor=basic_compressor.method_name, seed=args.seed)
    mine = shard_indices(len(dataset), rank, world)
    # the ingest mode follows from the WHOLE datalist -- the same decision on every rank, also on one whose shard is empty -- and is
    # settled before the process group forms, so that a datalist --ingest rows cannot take fails on all ranks together
    want = getattr(args, "ingest", "auto")
    all_bin = all(str(name).endswith(".bin") for name in dataset.data_list)
    if want == "rows" and not all_bin:
        raise SystemExit("--ingest rows needs a datalist of .bin files (float32 rows x, y, z, intensity)")
    ingest = "rows" if (len(dataset) > 0 and (want == "rows" or (want == "auto" and all_bin))) else "xyz"
    gather = None
    if getattr(args, "gather", False):
        import torch
        init_gather_group(rank, world, local)
        gdev = torch.device(device if os.environ.get("RPCC_DIST_BACKEND", "nccl") == "nccl" else "cpu")   # (gloo: CPU tensors)
        gather = RoundGather(len(dataset), rank, world, gdev, round_items=args.gather_round)
    t0 = time.time()
    stats = {"bytes": 0, "files": 0}
    with futures.ThreadPoolExecutor(args.workers) as pool:
        # staged, double-buffered feed (loader.StreamingCompressor): file reads of batch n+1, device work of batch n and
        # entropy coding + file output of batch n-1 overlap
        # .bin sweeps (float32 rows x, y, z, intensity: dataset/dataset.py:48-50) are read straight into the pinned staging slot
        # and go to the device as stored -- no np.fromfile + [:, :3] pass on the host; other formats are loaded and sliced
        sc = (streaming_factory or StreamingCompressor)(bc, batch=min(args.batch, max(len(mine), 1)), depth=4, workers=args.workers, pool=pool,
                                                        points_per_frame=getattr(args, "points_per_frame", None), ingest=ingest)
        names_of = {}

        def batches():
            for k, s in enumerate(range(0, len(mine), sc.B)):
                names = [dataset.data_list[i] for i in mine[s:s + sc.B]]
                names_of[k] = names
                frames = names if ingest == "rows" else list(pool.map(dataset.load_data, names))
                yield frames, [frame_identity(n) for n in names]

        def write_all(jobs):      # jobs: [(file name, bytes)]
            stats["bytes"] += sum(pool.map(lambda j: write_blob(args.output_dir, j[0], j[1]), jobs))
            stats["files"] += len(jobs)
            if args.output:
                for name, blob in jobs:
                    print("%s -> %d bytes" % (name, len(blob)))

        def sink(k, blobs):
            names = names_of.pop(k)
            if gather is None:
                write_all(list(zip(names, blobs)))
            else:     # the bytes go to rank 0, which writes them under their datalist names
                write_all([(dataset.data_list[i], blob) for i, blob in gather.add(blobs)])
        sc.run(_prefetch(batches()), sink=sink, entropy=True)
        if gather is not None:
            write_all([(dataset.data_list[i], blob) for i, blob in gather.finish()])
    dt = time.time() - t0
    if pinned is not None:
        print("rank %d/%d: pinned to CPUs %s" % (rank, world, ",".join(str(c) for c in pinned)))
    print("rank %d/%d: %d frames in %.3f s (%.1f frames/s incl. file I/O and entropy coding), %d files / %d bytes written%s"
          % (rank, world, len(mine), dt, len(mine) / max(dt, 1e-9), stats["files"], stats["bytes"],
             " (gathered to rank 0 over the process group)" if gather is not None else ""))
    if gather is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    a = make_parser(datalist=True).parse_args()
    print("Input arguments:")
    for key, val in vars(a).items():
        print("{:16} {}".format(key, val))
    compress(a)
