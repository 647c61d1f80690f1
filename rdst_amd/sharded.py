"""Multi-GPU route (Algorithm.GpuSharded): one process per GPU, torch.distributed over
RCCL/xGMI.  SURVEY.md §8(e):

  1. local top-level histogram (K1)                              -> counts[256]
  2. all_gather of the 256 counts of every rank                  -> counts[rank][digit]
     (north_star words this as an all-reduce of the global bucket counts; the all-gather is
     its superset and also yields the send/receive split tables)
  3. cut the 256 top digits into `world` contiguous ranges of near-equal global count
  4. one local stable scatter pass on the top digit (K3) groups the shard by owner rank
  5. all_to_all_single with the split sizes from step 2: every GPU ends up owning one
     contiguous key range
  6. local LSD sort of what arrived (all levels)

Rank r's return value is the r-th contiguous slice of the globally sorted array.  The only
collectives are one 2 KiB all-gather and one all-to-all-v of the keys: on the 8-GPU xGMI full
mesh the all-to-all drives all 7 links of every GPU at once.

The reference has no multi-device path; its nearest analogues are the MSD split +
par_bridge recursion of src/sorter.rs:131-138 and the tile->bucket regrouping of
src/sorts/recombinating_sort.rs:68-88.

The local steps go through an *engine*.  The product engine is `HipEngine` (the C ABI); tests
inject a CPU engine so that the exchange logic can run under gloo without a GPU.
"""
from typing import List, Optional, Sequence

import numpy as np

from . import radix_sort as _rs


class HipEngine:
    """Local steps on a HIP tensor through the C ABI (no fallback)."""

    def top_level_counts(self, keys) -> np.ndarray:
        levels = _rs.key_info(keys.dtype)[2]
        counts, _, _, _ = _rs.level_counts(keys, levels - 1)
        return np.asarray(counts, dtype=np.int64)

    def scatter_top_level(self, keys):
        return self.scatter_top_level_with_counts(keys)[0]

    def scatter_top_level_with_counts(self, keys):
        """One call: the scatter hook counts the level it moves (its K1), so the 256 top-level counts
        come with the grouped shard and no separate counting sweep is needed."""
        levels = _rs.key_info(keys.dtype)[2]
        dst, counts = _rs.scatter_level(keys, levels - 1)
        return dst, np.asarray(counts, dtype=np.int64)

    def sort(self, keys, tmp=None):
        _rs.sort_device_tensor(keys, tmp)
        return keys

    def empty(self, n, like):
        import torch
        return torch.empty(int(n), dtype=like.dtype, device=like.device)


def split_digits(global_counts: Sequence[int], world: int) -> List[int]:
    """Owner rank of each of the 256 top digits: contiguous ranges, each closed as soon as its
    running total reaches the ideal prefix (r+1)*N/world.  Deterministic, same on every rank."""
    total = int(sum(int(c) for c in global_counts))
    owner = [0] * 256
    r, run = 0, 0
    for d in range(256):
        owner[d] = r
        run += int(global_counts[d])
        while r < world - 1 and run * world >= (r + 1) * total and total > 0:
            r += 1
    return owner


def sharded_sort(local_keys, group=None, engine=None, return_info: bool = False):
    """Globally sort the concatenation of every rank's `local_keys`; returns this rank's slice
    (a new tensor whose length is the number of keys that fall into this rank's digit range).
    Collective: every rank of `group` must call it."""
    import torch
    import torch.distributed as dist

    if engine is None:
        engine = HipEngine()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        out = local_keys.clone()
        engine.sort(out)
        return (out, {"owner": [0] * 256, "recv": [out.numel()]}) if return_info else out

    dev = local_keys.device
    # RCCL ("nccl") moves device tensors directly.  Under gloo (CPU tests, or several test ranks
    # sharing one GPU) device tensors are staged through the host for the two collectives.
    via_host = local_keys.is_cuda and dist.get_backend(group) == "gloo"
    cdev = torch.device("cpu") if via_host else dev
    # 1 + 4. group my shard by top digit — one stable pass; owners are contiguous digit ranges, so this
    # is also the grouping by owner — and take the 256 counts the pass had to make anyway
    if hasattr(engine, "scatter_top_level_with_counts"):
        grouped, counts = engine.scatter_top_level_with_counts(local_keys)
    else:
        counts = engine.top_level_counts(local_keys)
        grouped = engine.scatter_top_level(local_keys)
    # 2. all-gather (2 KiB per rank)
    mine = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int64)).to(cdev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    table = torch.stack(gathered).cpu().numpy()  # [rank][digit]
    # 3. digit -> owner
    owner = np.asarray(split_digits(table.sum(axis=0), world))
    # send split: my keys per destination; receive split: every source's keys for my range
    send = [int(table[rank][owner == r].sum()) for r in range(world)]
    recv = [int(table[src][owner == rank].sum()) for src in range(world)]
    # 5. exchange
    inbox = engine.empty(sum(recv), local_keys)
    as_int = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[local_keys.element_size()]
    if via_host:
        host_in = torch.empty(sum(recv), dtype=as_int)
        dist.all_to_all_single(host_in, grouped.view(as_int).cpu(), output_split_sizes=recv, input_split_sizes=send,
                               group=group)
        inbox.view(as_int).copy_(host_in)
    else:
        dist.all_to_all_single(inbox.view(as_int), grouped.view(as_int), output_split_sizes=recv,
                               input_split_sizes=send, group=group)
    del grouped
    # 6. local LSD over every level (arrivals are only range-partitioned)
    engine.sort(inbox)
    if return_info:
        return inbox, {"owner": owner.tolist(), "recv": recv, "send": send}
    return inbox
