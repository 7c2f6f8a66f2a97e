"""Partition of work over ranks (one process per GPU): stereo pairs of a batch, or row bands of one large pair.

Pairs are independent, so the data path needs no collective: every rank processes its own
contiguous block of pairs; only the wall-clock of the slowest rank is reduced (MAX) to report
whole-job throughput."""


def pair_block(total_pairs, rank, world):
    """Contiguous block [first, first+count) of `total_pairs` owned by `rank`; blocks differ by at
    most one pair and cover every pair exactly once."""
    if world <= 0 or not (0 <= rank < world) or total_pairs < 0:
        raise ValueError("bad partition arguments")
    base, extra = divmod(total_pairs, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def pair_seed(pair_index, base_seed=0x5EED0001):
    """Seed of the synthetic pair with global index `pair_index` (seeds advance by 2: one for the
    texture, one for the noise)."""
    return (base_seed + 2 * pair_index) & 0xFFFFFFFF


def slowest_rank_seconds(elapsed, dist=None, device=None):
    """MAX over ranks of a per-rank elapsed time; identity without a process group."""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed)
    import torch
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_pair, pairs_done_all_ranks, seconds):
    """Whole-job rate in Mdisp/s."""
    return units_per_pair * pairs_done_all_ranks / seconds / 1e6


def row_band(rows, rank, world):
    """Rows [begin, end) of an image of `rows` rows owned by `rank`: contiguous bands that differ by at
    most one row and tile the image (same rule as pair_block)."""
    first, count = pair_block(rows, rank, world)
    return first, first + count


def gather_row_bands(band, rows, dist=None):
    """Assemble the full map from every rank's band.  `band` is this rank's (end - begin) x cols
    torch tensor; the result (on every rank) is rows x cols.  Bands differ in height by at most one
    row, so they are padded to a common height for all_gather and trimmed afterwards.  This is the
    one exchange step of the row-band scheme: outputs only, no halo traffic."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert band.shape[0] == rows
        return band
    world = dist.get_world_size()
    tall = max(row_band(rows, r, world)[1] - row_band(rows, r, world)[0] for r in range(world))
    padded = torch.zeros((tall,) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
    padded[:band.shape[0]] = band
    # the bytes are gathered: gloo has no collectives for 16-bit integers (disparity maps are int16)
    raw = padded.contiguous().view(torch.uint8)
    parts = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(parts, raw)
    out = []
    for r in range(world):
        b, e = row_band(rows, r, world)
        out.append(parts[r].view(band.dtype).reshape(padded.shape)[:e - b])
    return torch.cat(out, dim=0)


def merge_band_stats(cost_minmax, steer_max, dist=None, device="cpu"):
    """Element-wise min of the per-slice minima, max of the maxima and of the steer maxima over all ranks: the one
    (tiny) exchange of the row-band sharding.  cost_minmax: float32 [numDisp, 2]; returns the merged pair.  With
    dist=None (single process) the inputs are returned unchanged."""
    import numpy as np
    mm = np.ascontiguousarray(cost_minmax, dtype=np.float32)
    if dist is None:
        return mm, float(steer_max)
    import torch
    lo = torch.from_numpy(mm[:, 0].copy()).to(device)
    hi = torch.cat([torch.from_numpy(mm[:, 1].copy()).double(), torch.tensor([float(steer_max)], dtype=torch.float64)]).to(device)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    hi = hi.cpu().numpy()
    out = np.stack([lo.cpu().numpy(), hi[:-1].astype(np.float32)], axis=1)
    return out, float(hi[-1])


# ---- one pair over several devices: the rows of the SGM aggregation spread over the ranks --------------------------

def sgm_row_ranges(rows, band_rows, world):
    """Row range [begin, end) of each of `world` participants of a row-sharded SGM: whole bands of `band_rows` rows
    (mdisp_sgbm_rows_geometry), as even as the bands allow; a participant may come out empty when there are fewer
    bands than participants."""
    bands = (rows + band_rows - 1) // band_rows
    out = []
    for r in range(world):
        first, count = pair_block(bands, r, world)
        out.append((min(rows, first * band_rows), min(rows, (first + count) * band_rows)))
    return out


def check_row_ranges(ranges):
    """Raises ValueError when any participant's row range is empty (more participants than row bands).  The ranges are
    the same on every rank, so the error is raised everywhere at once instead of leaving some ranks blocked in a receive."""
    empty = [r for r, (b, e) in enumerate(ranges) if b >= e]
    if empty:
        raise ValueError("more participants than row bands: rank(s) %s would have no rows" % ", ".join(map(str, empty)))


def sgm_sweep_plan(rank, world):
    """The two sweeps of participant `rank`, in the order it runs them: [(direction, second, source, destination)].
    The downward wave-front (direction 0) starts at rank 0 and reaches rank r after r hops, the upward one (direction 1)
    starts at the last rank and reaches rank r after world - 1 - r hops; a participant runs first the sweep whose
    boundary states arrive first, so both wave-fronts cross the ranks at the same time and the pair takes the time of
    one sweep over the whole image instead of two.  source / destination: the rank the sweep's boundary states come
    from / go to (None at the image edge)."""
    if not (0 <= rank < world):
        raise ValueError("bad partition arguments")
    down = (0, rank - 1 if rank > 0 else None, rank + 1 if rank + 1 < world else None)
    up = (1, rank + 1 if rank + 1 < world else None, rank - 1 if rank > 0 else None)
    first, last = (down, up) if rank <= world - 1 - rank else (up, down)
    return [(first[0], 0, first[1], first[2]), (last[0], 1, last[1], last[2])]


def sgm_sequential_order(world):
    """An order in which ONE process can run the sweeps of all participants one after the other (every sweep after the
    one that produces its boundary states): [(rank, step)] with step indexing sgm_sweep_plan(rank, world)."""
    slots = []
    for r in range(world):
        for step, (direction, second, src, dst) in enumerate(sgm_sweep_plan(r, world)):
            hops = r if direction == 0 else world - 1 - r
            slots.append((2 * hops + second, r, step))      # a rank's second sweep never precedes its first
    done, order = set(), []
    pending = sorted(slots)
    while pending:
        for item in pending:
            _, r, step = item
            direction, second, src, dst = sgm_sweep_plan(r, world)[step]
            if (src is None or (src, direction) in done) and (step == 0 or (r, sgm_sweep_plan(r, world)[0][0]) in done):
                done.add((r, direction))
                order.append((r, step))
                pending.remove(item)
                break
        else:
            raise RuntimeError("sweep dependencies cannot be ordered")
    return order


def run_sweep_plan(rank, world, sweep, exchange, recv_buffer, before_sweep=None, after_sweep=None):
    """The exchange protocol of one participant: for each of its two sweeps, receive the boundary states the sweep starts
    from (unless it starts at the image edge), run `sweep(step, boundary_or_None)`, and pass the boundary states it
    returns on to the neighbour.  `before_sweep` / `after_sweep` order the sweep against the exchange (stream
    synchronisation on a GPU)."""
    for step, (direction, second, src, dst) in enumerate(sgm_sweep_plan(rank, world)):
        bnd = None
        if src is not None:
            exchange.recv(recv_buffer, src, direction)
            bnd = recv_buffer
        if before_sweep:
            before_sweep()
        out = sweep(step, bnd)
        if dst is not None:
            if after_sweep:
                after_sweep()
            exchange.send(out, dst, direction)
    exchange.wait()


class TorchBoundaryExchange:
    """Boundary states between neighbouring ranks over torch.distributed point-to-point calls: device tensors directly
    with the nccl (RCCL) backend, through host memory with gloo.  Sends do not block (both wave-fronts send towards
    each other at the same time); wait() completes them."""

    def __init__(self, dist, through_host=False):
        self.dist, self.through_host, self.pending = dist, through_host, []

    def send(self, tensor, dst, tag):
        t = tensor.cpu() if self.through_host else tensor
        self.pending.append((self.dist.isend(t, dst, tag=tag), t))

    def recv(self, tensor, src, tag):
        if self.through_host:
            import torch
            t = torch.empty(tensor.shape, dtype=tensor.dtype)
            self.dist.recv(t, src, tag=tag)
            tensor.copy_(t)
        else:
            self.dist.recv(tensor, src, tag=tag)

    def wait(self):
        for w, _ in self.pending:
            w.wait()
        self.pending = []


class RowShardedSGBM:
    """One participant of a pair whose rows are spread over `world` devices (mdisp_sgbm_rows_sweep_dev).

    compute() runs the participant's two sweeps and writes its rows of the disparity map; the other rows of `d_disp` are
    left alone (gather_row_bands / an all_gather of the row ranges assembles the map).  The participants' maps together
    equal the map of mdisp_sgbm_dev bit for bit."""

    def __init__(self, ctx, rows, cols, params, rank, world, device="cuda"):
        import ctypes as C
        import torch
        from . import _lib
        self.lib, self.ctx, self.params = _lib.load(), ctx, params
        self.rows, self.cols, self.rank, self.world = rows, cols, rank, world
        band, bnd, ws = C.c_int(), C.c_size_t(), C.c_size_t()
        _lib.check(self.lib.mdisp_sgbm_rows_geometry(rows, cols, C.byref(params), C.byref(band), C.byref(bnd), C.byref(ws)))
        self.band_rows, self.boundary_bytes, self.ws_bytes = band.value, bnd.value, ws.value
        ranges = sgm_row_ranges(rows, band.value, world)
        # every rank computes the same ranges, so every rank raises together: raising only on the empty ranks would
        # leave their neighbours waiting for boundary states that never come
        check_row_ranges(ranges)
        self.begin, self.end = ranges[rank]
        self.ws = torch.empty(ws.value, dtype=torch.uint8, device=device)
        self.bnd_in = torch.empty(bnd.value, dtype=torch.uint8, device=device)
        self.bnd_out = [torch.empty(bnd.value, dtype=torch.uint8, device=device) for _ in range(2)]

    def sweep(self, step, d_left, d_right, pitch, d_disp, boundary_in):
        """Sweep `step` (0 or 1) of this participant's plan; returns the boundary-state tensor for the neighbour (or
        None).  Asynchronous on the context's stream."""
        import ctypes as C
        from . import _lib
        direction, second, src, dst = sgm_sweep_plan(self.rank, self.world)[step]
        out = self.bnd_out[step] if dst is not None else None
        _lib.check(self.lib.mdisp_sgbm_rows_sweep_dev(
            self.ctx.handle, d_left.data_ptr(), d_right.data_ptr(), self.rows, self.cols, pitch, C.byref(self.params),
            self.begin, self.end, direction, second, boundary_in.data_ptr() if boundary_in is not None else None,
            out.data_ptr() if out is not None else None, d_disp.data_ptr(), self.ws.data_ptr(), self.ws_bytes))
        return out

    def compute(self, d_left, d_right, pitch, d_disp, exchange):
        """Both sweeps with the boundary exchange in between; `exchange` has send(tensor, dst, tag), recv(tensor, src,
        tag) and wait().  Synchronises the context's stream (a hand-off timeout raises here)."""
        import torch
        run_sweep_plan(self.rank, self.world, lambda step, bnd: self.sweep(step, d_left, d_right, pitch, d_disp, bnd), exchange, self.bnd_in,
                       before_sweep=torch.cuda.synchronize,      # the receive ran on torch's stream, the sweep runs on the context's
                       after_sweep=self.ctx.synchronize)
        self.ctx.synchronize()
