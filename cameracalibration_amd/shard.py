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
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    out = []
    for r in range(world):
        b, e = row_band(rows, r, world)
        out.append(parts[r][:e - b])
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
