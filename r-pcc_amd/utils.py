"""Config loading (mirror of the reference's utils/utils.py:18-25)."""
import os
import zlib

import yaml


class Config(dict):
    """dict with attribute access (what the reference gets from EasyDict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__


def load_yaml(path):
    with open(path, "r") as f:
        return Config(yaml.safe_load(f))


def load_compressor_cfg(yaml_file):
    """utils/utils.py:18-25: YAML -> attribute dict."""
    return load_yaml(yaml_file)


def frame_identity(path):
    """Stable identity of a frame for the seeded RANSACs of this build (rpcc_ground_ransac / rpcc_plane_model:
    frame_ids): CRC-32 of the file's base name.  A file's planes -- and so its .rpcc bytes -- then do not depend on the
    batch size, the position in the batch, the number of ranks or the tool (compress.py / compress_datalist.py) that
    processes it.  (The reference's Open3D RANSAC is unseeded: its output is not reproducible at all.)"""
    return zlib.crc32(os.path.basename(str(path).strip()).encode()) & 0x7FFFFFFF


def available_cpus():
    """CPUs this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota (cpu.max) -- a
    container that shows 256 logical CPUs may be limited to 16 of them, and 256 threads then only time-slice."""
    import math
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:   # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, math.ceil(q / p)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def visible_gpus():
    """Number of GPUs this process would see, WITHOUT initialising HIP / HSA here (bench.py's parent counts before it
    spawns its rank processes and must stay a process that never touched the GPU): the KFD topology in sysfs (nodes with
    simd_count > 0 are GPUs), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one of them
    is a plain index list.  None when sysfs tells nothing (not a ROCm host)."""
    import glob
    import re
    n = None
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        n = 0
        for p in nodes:
            try:
                m = re.search(r"^simd_count\s+(\d+)", open(p).read(), re.M)
            except OSError:
                continue
            if m and int(m.group(1)) > 0:
                n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is None:
            continue
        items = [s for s in v.split(",") if s.strip() != ""]
        if all(re.fullmatch(r"\s*\d+\s*", s) for s in items):
            n = len(items) if n is None else min(n, len(items))
        elif v.strip() == "":
            n = 0
    return n


def log_affinity_skipped(why):
    import sys
    print("rpcc_amd: CPU pinning skipped: " + why, file=sys.stderr)


def local_rank_env():
    """(LOCAL_RANK, LOCAL_WORLD_SIZE) when the launcher set BOTH (torchrun does), else None: without them a multi-node job would size the
    slices by the global world and pin every rank of a host to the same first slice."""
    lr, lw = os.environ.get("LOCAL_RANK"), os.environ.get("LOCAL_WORLD_SIZE")
    if lr is None or lw is None:
        return None
    try:
        return int(lr), int(lw)
    except ValueError:
        return None


def pin_rank_cpus(local_rank, local_world):
    """One process per GPU on one host: rank r keeps the r-th of `local_world` equal, contiguous slices of the CPUs the job may use, so
    that the ranks' feeder threads (file reads, staging copies, entropy coding: loader.StreamingCompressor) do not migrate over each other's
    cores and caches.  -> the CPUs kept (sorted), or None when nothing was changed (single rank, RPCC_NO_AFFINITY=1, fewer CPUs than
    ranks, or a platform without sched_setaffinity)."""
    if local_world <= 1 or os.environ.get("RPCC_NO_AFFINITY") == "1":
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None
    # A launcher that has already bound this rank to its own CPU set (numactl, a scheduler's cgroup cpuset per task) must not be cut again:
    # the slices below assume that every rank starts from the same host-wide mask.  A mask smaller than the machine says it has been.
    if len(cpus) < (os.cpu_count() or len(cpus)) and os.environ.get("RPCC_FORCE_AFFINITY") != "1":
        log_affinity_skipped("the process already runs on %d of %d CPUs (bound by its launcher)" % (len(cpus), os.cpu_count()))
        return None
    if not 0 <= local_rank < local_world:
        log_affinity_skipped("LOCAL_RANK %d outside LOCAL_WORLD_SIZE %d" % (local_rank, local_world))
        return None
    per = len(cpus) // local_world
    if per < 1:
        return None
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine
