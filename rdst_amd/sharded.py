"""Multi-GPU route (Algorithm.GpuSharded): one process per GPU, torch.distributed over
RCCL/xGMI.  SURVEY.md §8(e):

  1. one local stable scatter pass on the top digit (K3, its own K1 counting that level only) groups the
     shard by top digit — owners are contiguous digit ranges, so this is the grouping by owner — and
     leaves the 256 digit counts ON THE DEVICE                  (rdst_hip_split_top_level_device, non-blocking)
  2. all_gather of the 256 counts of every rank                  -> counts[rank][digit], a device tensor
     (north_star words this as an all-reduce of the global bucket counts; the all-gather is its superset
     and also yields the send / receive split tables)
  3. on the device: cut the 256 top digits into `world` contiguous ranges of near-equal global count, and
     from that the send and receive splits; ONE small device-to-host copy brings the 2*world split sizes
     (torch's all_to_all_single takes them as host integers)
  4. all_to_all_single with those splits: every GPU ends up owning one contiguous key range
  5. local sort of what arrived (all levels; the device picks its route as for any slice)

Skewed top bytes (last paragraph): when one top digit holds more than a rank's fair share
the digit ranges cannot balance, and the split is redone on the top SIXTEEN bits: two stable passes order
the shard by them (rdst_hip_split_top16_device), the 65 536 bucket lengths are all-gathered, owners become
contiguous ranges of 16-bit prefixes.  Correct for any input either way; 16 bits balance whatever 65 536
buckets can balance.

Rank r's return value is the r-th contiguous slice of the globally sorted array.  The only collectives are
one small all-gather and one all-to-all-v of the keys: on the 8-GPU xGMI full mesh the all-to-all drives all
7 links of every GPU at once.

The reference has no multi-device path; its nearest analogues are the MSD split + par_bridge recursion of
src/sorter.rs:131-138 and the tile->bucket regrouping of src/sorts/recombinating_sort.rs:68-88.

The local steps go through an *engine*.  The product engine is `HipEngine` (the C ABI); tests inject a CPU
engine so that the exchange logic can run under gloo without a GPU.
"""
import ctypes
from typing import List, Sequence

import numpy as np

from . import _lib
from . import radix_sort as _rs

# a rank may receive this many times its fair share before the 16-bit split is tried
SKEW_SLACK = 1.25


class HipEngine:
    """Local steps on a HIP tensor through the C ABI (no fallback).  Everything is enqueued on the tensor's
    current stream and nothing here blocks."""

    def split_top_level(self, keys):
        """(shard grouped by top digit — a new tensor —, int64 device tensor of the 256 digit counts)"""
        import torch
        kind, nbytes, _levels = _rs.key_info(keys.dtype)
        dst = torch.empty_like(keys)
        counts = torch.empty(256, dtype=torch.int64, device=keys.device)
        with torch.cuda.device(keys.device):
            _lib.check(_lib.load().rdst_hip_split_top_level_device(
                ctypes.c_void_p(keys.data_ptr()), ctypes.c_void_p(dst.data_ptr()), keys.numel(), nbytes, kind,
                ctypes.c_void_p(counts.data_ptr()), _rs._stream_handle(keys)))
        return dst, counts

    def split_top16(self, keys):
        """(shard ordered by the top 16 bits of the mapped key — a new tensor —, int64 device tensor of the 65 536
        bucket lengths)"""
        import torch
        kind, nbytes, _levels = _rs.key_info(keys.dtype)
        work = keys.clone()
        tmp = torch.empty_like(keys)
        counts = torch.empty(65536, dtype=torch.int64, device=keys.device)
        with torch.cuda.device(keys.device):
            _lib.check(_lib.load().rdst_hip_split_top16_device(
                ctypes.c_void_p(work.data_ptr()), ctypes.c_void_p(tmp.data_ptr()), keys.numel(), nbytes, kind,
                ctypes.c_void_p(counts.data_ptr()), _rs._stream_handle(keys)))
        return work, counts

    def sort(self, keys, tmp=None):
        _rs.sort_device_tensor(keys, tmp, check=False)
        return keys

    def empty(self, n, like):
        import torch
        return torch.empty(int(n), dtype=like.dtype, device=like.device)


def split_digits(global_counts: Sequence[int], world: int) -> List[int]:
    """Owner rank of each bucket (256 top digits, or 65 536 prefixes): contiguous ranges; a bucket goes to the rank whose
    ideal range [r*N/world, (r+1)*N/world) holds its MIDDLE key.  Deterministic, same on every rank.  Host form of `_owners`
    below (kept for the tests and as the definition).  (The middle, not the first key: on uniform keys the ranges then end
    exactly on multiples of 256/world digits — every rank's keys share their top log2(world) bits, which the local sort's
    atomic route turns into a lowered bucket window — where "first key" hands rank r one digit of rank r+1's every time a
    running total falls a hair short.)"""
    total = int(sum(int(c) for c in global_counts))
    owner = [0] * len(global_counts)
    below = 0
    for d in range(len(global_counts)):
        c = int(global_counts[d])
        owner[d] = min(world - 1, (world * (2 * below + c)) // (2 * total)) if total > 0 else 0
        below += c
    return owner


def _owners(table, world):
    """split_digits on the device: owner[d] = min(world-1, floor(world * (keys below d + half of d's) / total))."""
    import torch
    glob = table.sum(dim=0)
    total = glob.sum()
    below = torch.cumsum(glob, dim=0) - glob
    owner = torch.div((2 * below + glob) * world, torch.clamp(2 * total, min=1), rounding_mode="floor")
    return torch.clamp(owner, max=world - 1)


def _splits(table, owner, rank, world):
    """(send[world], recv[world], largest share any rank receives, total) as ONE int64 device tensor of 2*world+2"""
    import torch
    ranks = torch.arange(world, device=table.device, dtype=owner.dtype)
    mine = (owner[None, :] == ranks[:, None]).to(table.dtype)          # [world, buckets]: bucket belongs to rank r
    send = (table[rank][None, :] * mine).sum(dim=1)                    # my keys per destination
    recv = (table * mine[rank][None, :]).sum(dim=1)                    # every source's keys for my range
    share = (table.sum(dim=0)[None, :] * mine).sum(dim=1)              # what each rank will own
    return torch.cat([send, recv, share.max()[None], table.sum()[None]])


def sharded_sort(local_keys, group=None, engine=None, return_info: bool = False, force_collectives: bool = False, timings=None):
    """Globally sort the concatenation of every rank's `local_keys`; returns this rank's slice
    (a new tensor whose length is the number of keys that fall into this rank's range).
    Collective: every rank of `group` must call it.  `timings` (a dict, device tensors only): filled with the milliseconds this
    rank spent per stage — split, counts + plan, exchange, local sort — from events on the current stream; the call then ends
    with a synchronisation (a diagnostic step, not the timed path)."""
    import torch
    import torch.distributed as dist

    marks = []

    def mark(stage):
        if timings is not None and local_keys.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((stage, ev))

    def close_marks():
        if marks:
            torch.cuda.synchronize()
            for (_, a), (stage, b) in zip(marks[:-1], marks[1:]):
                timings[stage] = timings.get(stage, 0.0) + a.elapsed_time(b)

    if engine is None:
        engine = HipEngine()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not force_collectives:  # (force_collectives: a one-rank group still runs every collective — the RCCL smoke test)
        out = local_keys.clone()
        engine.sort(out)
        return (out, {"owner": [0] * 256, "recv": [out.numel()], "send": [out.numel()], "split_bits": 8}) if return_info else out

    # RCCL ("nccl") moves device tensors directly.  Under gloo (CPU tests, or several test ranks
    # sharing one GPU) device tensors are staged through the host for the two collectives.
    via_host = local_keys.is_cuda and dist.get_backend(group) == "gloo"

    def gather_table(counts):
        mine = counts.cpu() if via_host else counts
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        return torch.stack(gathered).to(counts.device)                 # [rank][bucket], where the counts live

    # 1. + 2.  top-digit split of my shard, counts stay on the device; all-gather of the 256 counts
    mark("start")
    grouped, counts = engine.split_top_level(local_keys)
    mark("split")
    table = gather_table(counts)
    # 3.  digit -> owner, split sizes: device arithmetic, one small copy to the host
    owner = _owners(table, world)
    plan = _splits(table, owner, rank, world).cpu().tolist()           # the one synchronising device-to-host copy
    split_bits = 8
    total, worst = plan[2 * world + 1], plan[2 * world]
    if total > 0 and worst * world > SKEW_SLACK * total and local_keys.element_size() >= 2:
        # one top digit holds too much: split on the top 16 bits instead (two passes; the fallback, not the rule)
        del grouped
        grouped, counts16 = engine.split_top16(local_keys)
        table = gather_table(counts16)
        owner = _owners(table, world)
        plan = _splits(table, owner, rank, world).cpu().tolist()
        split_bits = 16
    send = [int(x) for x in plan[:world]]
    recv = [int(x) for x in plan[world:2 * world]]
    mark("counts_and_plan")
    # 4.  exchange
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
    mark("exchange")
    # 5.  local sort over every level (arrivals are only range-partitioned)
    engine.sort(inbox)
    mark("local_sort")
    close_marks()
    if return_info:
        return inbox, {"owner": owner.cpu().tolist(), "recv": recv, "send": send, "split_bits": split_bits}
    return inbox
