"""Sharding of independent buffers over the GPUs of one node (BASELINE.json configs 4/5).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, or "gloo" in the CPU
tests).  Buffers are independent units: every rank compresses / decompresses the buffers it
owns with no data-path collective; the only exchange is the result gather to one rank:

    sizes    all_reduce(SUM) of an int64[count] vector in which each rank filled its own entries
    payload  one gather of the per-rank concatenations, padded to the largest rank total

The reference has no counterpart (it is single-threaded: README.md:28-42 shows a caller's loop);
the per-buffer semantics (result bytes, thrown error) are exactly those of deflate()/inflate().
"""
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def partition(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Size-balanced, deterministic owner lists: longest buffers first onto the lightest rank."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    owned: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owned[r].append(i)
        load[r] += int(sizes[i])
    for o in owned:
        o.sort()
    return owned


def run_sharded(buffers: Sequence[np.ndarray], engine: Callable[[np.ndarray], Tuple[int, np.ndarray]], group=None,
                dst: int = 0, device=None) -> Optional[List[Tuple[int, np.ndarray]]]:
    """Runs `engine` (buffer -> (status, bytes)) on this rank's share and gathers everything on `dst`.

    Every rank passes the same `buffers` list (only the owned ones are touched).  Returns, on
    `dst`, a list of (status, bytes) in the original order; None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    count = len(buffers)
    owned = partition([len(b) for b in buffers], world)
    mine = owned[rank]
    results = {i: engine(buffers[i]) for i in mine}

    dev = device if device is not None else torch.device("cpu")
    sizes = torch.zeros(count, dtype=torch.int64, device=dev)
    status = torch.zeros(count, dtype=torch.int64, device=dev)
    for i, (st, data) in results.items():
        sizes[i] = len(data)
        status[i] = st
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(status, op=dist.ReduceOp.SUM, group=group)
    sizes_h = sizes.cpu().numpy()
    totals = [int(sum(int(sizes_h[i]) for i in owned[r])) for r in range(world)]
    pad = max(max(totals), 1)
    local = np.zeros(pad, dtype=np.uint8)
    pos = 0
    for i in mine:
        d = results[i][1]
        local[pos:pos + len(d)] = d
        pos += len(d)
    local_t = torch.from_numpy(local).to(dev)
    gathered = [torch.empty(pad, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(local_t, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    out: List[Optional[Tuple[int, np.ndarray]]] = [None] * count
    status_h = status.cpu().numpy()
    for r in range(world):
        blob = gathered[r].cpu().numpy()
        pos = 0
        for i in owned[r]:
            n = int(sizes_h[i])
            out[i] = (int(status_h[i]), blob[pos:pos + n].copy())
            pos += n
    return out  # type: ignore[return-value]


def gpu_engines(z, device):
    """(deflate_engine, inflate_engine) running one buffer through the HBM-resident C-ABI entry points."""
    import torch

    def run_deflate(buf: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(buf)).to(device)
        try:
            return 0, z.deflate_tensor(t).cpu().numpy()
        except z.ZlibEsError as e:
            return e.code, np.zeros(0, dtype=np.uint8)

    def run_inflate(buf: np.ndarray):
        try:
            return 0, z.inflate(buf)
        except z.ZlibEsError as e:
            return e.code, np.zeros(0, dtype=np.uint8)

    return run_deflate, run_inflate
