"""Window sharding across the GPUs of one node: windows are independent, so each rank takes a contiguous
range and there is no collective on the data path; results are gathered in window order on rank 0 only when a
caller wants one ordered output (the drivers do; bench.py does not)."""


def shard_range(n_items, rank, world):
    """Contiguous, balanced, order-preserving split: rank r gets [lo, hi)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_by_cost(costs, world):
    """Contiguous split of a cost list into `world` ranges with near-equal total cost (piles differ in depth).
    Returns a list of (lo, hi).  Order-preserving, deterministic, covers everything exactly once."""
    n = len(costs)
    total = float(sum(costs))
    out, lo, acc = [], 0, 0.0
    for r in range(world):
        if r == world - 1:
            out.append((lo, n))
            break
        target = total * (r + 1) / world
        hi = lo
        while hi < n - (world - 1 - r) and acc + costs[hi] <= target:
            acc += costs[hi]
            hi += 1
        if hi == lo and lo < n - (world - 1 - r):
            acc += costs[hi]
            hi += 1
        out.append((lo, hi))
        lo = hi
    return out


def gather_in_order(local_items, rank, world, dist=None):
    """Gather per-rank lists on rank 0 in rank order (== window order for contiguous shards)."""
    if world == 1 or dist is None:
        return list(local_items)
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(list(local_items), bucket, dst=0)
    if rank != 0:
        return None
    out = []
    for part in bucket:
        out.extend(part)
    return out
