"""Partition of a batch of stereo pairs over ranks (one process per GPU).

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
