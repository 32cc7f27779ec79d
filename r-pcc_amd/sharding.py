"""Frame sharding and payload gathering for one-process-per-GPU runs.

Frames are independent, so the data path has no collective: rank r of R compresses datalist entries
r, r+R, r+2R, ...  The only exchange is the final gather of the variable-length payloads to rank 0:
an all_gather of the byte counts, then a gather of the padded byte buffers (RCCL on GPUs, gloo in the
CPU tests).  Both functions take the process group as an argument and work with any backend."""
import math

import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """Round-robin shard of range(n_items) for `rank` of `world`."""
    return list(range(rank, n_items, world))


def owner_of(index, world):
    return index % world


def gather_payloads(local_payloads, device, group=None, dst=0):
    """local_payloads: list of `bytes`/uint8 tensors produced by this rank (its shard, in shard order).
    Returns on `dst` the payloads of ALL ranks re-interleaved into datalist order, elsewhere None.
    Two collectives: all_gather of the per-frame sizes, gather of the concatenated bytes (padded to the
    largest rank total)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    tens = [p if torch.is_tensor(p) else torch.frombuffer(bytearray(p), dtype=torch.uint8) for p in local_payloads]
    sizes = torch.tensor([t.numel() for t in tens], dtype=torch.int64, device=device)
    n_local = torch.tensor([sizes.numel()], dtype=torch.int64, device=device)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    max_n = max(counts) if counts else 0
    pad_sizes = torch.zeros(max_n, dtype=torch.int64, device=device)
    pad_sizes[: sizes.numel()] = sizes
    all_sizes = [torch.zeros(max_n, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, pad_sizes, group=group)
    totals = [int(s[:c].sum().item()) for s, c in zip(all_sizes, counts)]
    max_bytes = max(totals) if totals else 0
    buf = torch.zeros(max(max_bytes, 1), dtype=torch.uint8, device=device)
    if tens:
        cat = torch.cat([t.to(device) for t in tens]) if len(tens) > 1 else tens[0].to(device)
        buf[: cat.numel()] = cat
    recv = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst, group=group)
    if rank != dst:
        return None
    per_rank = []
    for r in range(world):
        off, items = 0, []
        for k in range(counts[r]):
            n = int(all_sizes[r][k].item())
            items.append(bytes(recv[r][off:off + n].cpu().numpy().tobytes()))
            off += n
        per_rank.append(items)
    total = sum(counts)
    out = [None] * total
    for r in range(world):
        for k, item in enumerate(per_rank[r]):
            out[r + k * world] = item
    return out


class RoundGather:
    """The north-star exchange of the datalist driver (tools/compress_datalist.py --gather): every rank entropy-codes its
    round-robin shard; the finished .rpcc byte strings travel to rank `dst` in rounds of `round_items` frames per rank
    (gather_payloads: all_gather of the sizes + one padded gather of the bytes) and come out there in DATALIST order with
    their datalist index.  Every rank calls the same number of rounds -- ceil(ceil(n / world) / round_items), fixed by the
    datalist length alone -- a rank whose shard is shorter sends what it has."""

    def __init__(self, n_items, rank, world, device, round_items=4096, group=None, dst=0):
        self.n, self.rank, self.world, self.device, self.group, self.dst = int(n_items), rank, world, device, group, dst
        self.R = max(1, int(round_items))
        self.mine = len(shard_indices(self.n, rank, world))
        longest = len(shard_indices(self.n, 0, world))
        self.rounds = (longest + self.R - 1) // self.R
        self.done_rounds = 0
        self.pending = []          # this rank's blobs of the current and later rounds, in shard order
        self.taken = 0             # blobs of this rank already sent

    def _round(self):
        j = self.done_rounds
        want = max(0, min(self.R, self.mine - j * self.R))       # this rank's share of round j
        assert len(self.pending) >= want
        send, self.pending = self.pending[:want], self.pending[want:]
        out = gather_payloads(send, self.device, group=self.group, dst=self.dst)
        self.done_rounds += 1
        self.taken += want
        if out is None:
            return []
        base = j * self.R * self.world       # first datalist index of the round: local item k of rank r is entry r + k * world
        return [(base + i, blob) for i, blob in enumerate(out)]

    def add(self, blobs):
        """blobs: the next finished payloads of this rank, in shard order.  Runs every round that is complete on this rank;
        returns [(datalist index, bytes)] on dst (datalist order inside a round), [] elsewhere."""
        self.pending.extend(blobs)
        out = []
        while self.done_rounds < self.rounds and (len(self.pending) >= self.R or self.taken + len(self.pending) >= self.mine):
            out.extend(self._round())
        return out

    def finish(self):
        """Call once after the last add(): runs the rounds that are still open (ranks without items left send nothing)."""
        out = []
        while self.done_rounds < self.rounds:
            out.extend(self._round())
        assert not self.pending, "more payloads handed in than the shard holds"
        return out


class PackedExchange:
    """The per-step exchange of bench.py / the batch drivers.  Every step the ranks all_gather the per-frame
    payload lengths nnz (what rank `dst` needs to index the job; B * 4 bytes per rank).  With payloads=True every rank also
    sends its frames' residual stream, packed back to back (rpcc_pack_payload; int16 entries, at most `cap` of them, cap =
    the largest rank's point count, agreed on once with agree_capacity), to `dst` as BYTES (RCCL has no 16-bit integer
    type); with payloads=False the payload stays with the rank that produced it (each rank writes its own .rpcc files,
    tools/compress_datalist.py).  The receive buffers are allocated once."""

    def __init__(self, frames_per_rank, cap, device, group=None, dst=0, payloads=True):
        self.world, self.rank, self.group, self.dst, self.cap = dist.get_world_size(group), dist.get_rank(group), group, dst, int(cap)
        self.payloads = bool(payloads)
        self.nnz_all = [torch.empty((frames_per_rank,), dtype=torch.int32, device=device) for _ in range(self.world)]
        self.pay_all = ([torch.empty((2 * self.cap,), dtype=torch.uint8, device=device) for _ in range(self.world)]
                        if self.rank == dst and self.payloads else None)

    @staticmethod
    def agree_capacity(local_points, device, group=None):
        t = torch.tensor([int(local_points)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return int(t.item())

    def bytes_per_step(self):
        """Bytes this exchange moves per step over all ranks (lengths to everyone, payloads to dst)."""
        n = self.nnz_all[0].numel() * 4 * self.world
        return n + (2 * self.cap * (self.world - 1) if self.payloads else 0)

    def step(self, packed_i16, nnz_i32):
        """packed_i16: int16 [cap] (this rank's stream, frames back to back; ignored without payloads),
        nnz_i32: int32 [frames_per_rank]."""
        dist.all_gather(self.nnz_all, nnz_i32, group=self.group)
        if self.payloads:
            assert packed_i16.dtype == torch.int16 and packed_i16.numel() == self.cap
            dist.gather(packed_i16.view(torch.uint8), self.pay_all, dst=self.dst, group=self.group)

    def frame_stream(self, rank, frame):
        """On dst, after step() completed: the int16 residual run of `frame` of `rank` (a view into the receive buffer)."""
        nnz = self.nnz_all[rank].to(torch.int64)
        start = int(nnz[:frame].sum().item())
        return self.pay_all[rank].view(torch.int16)[start:start + int(nnz[frame].item())]


def agree_steps(requested, warm_step_s, min_region_s, device, group=None, cap=100000):
    """Length of a multi-rank timed region (bench.py): at least `requested` steps and at least `min_region_s` seconds by the
    slowest-to-decide rank's own warm-up step time; every rank returns the same number (the ranks run a collective per step)."""
    want = int(requested)
    if warm_step_s and warm_step_s > 0:
        want = max(want, min(int(math.ceil(min_region_s / warm_step_s - 1e-9)), int(cap)))
    t = torch.tensor([want], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def gather_rank_times(dt_local, device, group=None):
    """Every rank's own time of a timed region, in rank order, on every rank (the job's time is the maximum)."""
    world = dist.get_world_size(group)
    allt = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(allt, torch.tensor([float(dt_local)], dtype=torch.float64, device=device), group=group)
    return [float(t.item()) for t in allt]
