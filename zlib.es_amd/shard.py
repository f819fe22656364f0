"""Sharding of independent buffers over the GPUs of one node (BASELINE.json configs 3/4).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, or "gloo" in the CPU
tests).  Buffers are independent units: every rank compresses / decompresses the buffers it
owns with no data-path collective; the only exchange is the result gather to one rank:

    sizes    all_reduce(SUM) of an int64[2 * count] vector (length, status) in which each rank filled
             the entries of the buffers it owns
    payload  every rank sends ONE message — the concatenation of its results, exact length — to the
             destination rank, which posts the matching receives into one arena (grouped send/recv:
             RCCL has no variable-length gather, and a padded `gather` would move the largest rank
             total world times)

Payloads stay where they are produced: on a GPU run the results are device tensors from the C-ABI's
device entry points, the messages leave HBM over xGMI and land in the destination's HBM; nothing goes
through host memory.  xGMI is point to point: the destination receives over one link per peer at once.

The reference has no counterpart (it is single-threaded: README.md:28-42 shows a caller's loop);
the per-buffer semantics (result bytes, thrown error) are exactly those of deflate()/inflate().

Single-buffer deflate over several GPUs is `deflate_split` below: blocks of the
reference format are independent (src/deflate.ts:20-34, src/lz77.ts:11-22), so contiguous block
ranges go to the ranks, and the seams are closed from three numbers per rank.
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


class Gathered:
    """What the destination rank holds after a gather: one arena with every buffer's result.

    arena   uint8 tensor (on the destination's device)
    offset  offset[i], length[i] — result i is arena[offset[i] : offset[i] + length[i]]
    status  status[i] — 0 or the zes_status the per-buffer call returned (its bytes are then empty)
    """

    def __init__(self, arena, offset, length, status):
        self.arena, self.offset, self.length, self.status = arena, offset, length, status

    def result(self, i):
        return self.arena[self.offset[i]: self.offset[i] + self.length[i]]


class PendingGather:
    """A gather whose transfers are in flight.  `finish()` completes it: returns the Gathered on `dst`, None
    elsewhere.  Nothing in `gather_begin` waits for the device or for a peer — the host learns the sizes here."""

    def __init__(self, fin):
        self._fin, self._out, self._done = fin, None, False

    def finish(self):
        if not self._done:
            self._out, self._done, self._fin = self._fin(), True, None
        return self._out

    wait = finish


def _slack(nbytes: int) -> int:
    """Room kept behind a rank's hinted total in the destination's arena: results may grow by this much against
    the hint before the slow path (a second, exact-length gather) is taken."""
    return max(nbytes >> 6, 4096)


def gather_begin(local, my_ids: Sequence[int], my_len: Sequence[int], my_status: Sequence[int], owned: List[List[int]],
                 count: int, group=None, dst: int = 0, hint: Optional[Sequence[int]] = None) -> PendingGather:
    """Starts the exchange step and returns at once.  `local` = this rank's results back to back, in the order of
    `my_ids` (a uint8 tensor on this rank's device; for "nccl" a CUDA tensor, and it never leaves the device).

    Without `hint` the receiver has to learn the lengths before it can post exact-length receives: the
    (length, status) all_reduce is read on the host first (one wait for that small collective).

    With `hint` — the per-buffer lengths every rank expects, e.g. `Gathered.length` / `PendingGather.lengths` of
    the step before — nothing waits: the all_reduce is started, and a first round of messages is posted from the
    hint alone, rank r -> dst exactly H[r] = sum of its hinted lengths (a sender whose results came out shorter
    pads the message, one whose results came out longer sends the rest in a second round).  `finish()` reads the
    all_reduce — by then behind the transfers — and posts that second round where needed: both sides decide from
    the same table, so sends and receives always pair up with equal sizes.  A rank whose results outgrew its
    arena segment (hint + slack) makes every rank repeat the gather the exact-length way.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local.device
    meta = torch.zeros(2 * count, dtype=torch.int64, device=dev)
    if len(my_ids):
        ids = torch.as_tensor(list(my_ids), dtype=torch.int64, device=dev)
        meta[ids] = torch.as_tensor(list(my_len), dtype=torch.int64, device=dev)
        meta[ids + count] = torch.as_tensor(list(my_status), dtype=torch.int64, device=dev)
    my_total = int(sum(int(x) for x in my_len))
    assert my_total == local.numel(), (my_total, local.numel())

    def read_meta(work):
        work.wait()
        meta_h = meta.cpu().numpy()
        return [int(x) for x in meta_h[:count]], [int(x) for x in meta_h[count:]]

    def exact(length, status):
        """The exact-length round: every rank's whole result in one message."""
        totals = [sum(length[i] for i in owned[r]) for r in range(world)]
        ops, out = [], None
        if rank == dst:
            base, pos = [0] * world, 0
            for r in range(world):
                base[r] = pos
                pos += totals[r]
            arena = torch.empty(max(pos, 1), dtype=torch.uint8, device=dev)
            offset = [0] * count
            for r in range(world):
                p = base[r]
                for i in owned[r]:
                    offset[i] = p
                    p += length[i]
                if r == rank:
                    arena[base[r]: base[r] + totals[r]].copy_(local)
                elif totals[r]:
                    ops.append(dist.P2POp(dist.irecv, arena[base[r]: base[r] + totals[r]], r, group=group))
            out = Gathered(arena, offset, length, status)
        elif totals[rank]:
            ops.append(dist.P2POp(dist.isend, local, dst, group=group))
        return out, (dist.batch_isend_irecv(ops) if ops else [])

    work_meta = dist.all_reduce(meta, op=dist.ReduceOp.SUM, group=group, async_op=True)
    if hint is None:
        length, status = read_meta(work_meta)
        out, works = exact(length, status)

        def fin_plain():
            for w in works:
                w.wait()
            return out

        pg = PendingGather(fin_plain)
        pg.lengths = length
        return pg

    hint = [int(x) for x in hint]
    assert len(hint) == count
    H = [sum(hint[i] for i in owned[r]) for r in range(world)]
    cap = [H[r] + _slack(H[r]) for r in range(world)]
    base, pos = [0] * world, 0
    for r in range(world):
        base[r] = pos
        pos += (cap[r] + 15) // 16 * 16
    ops, arena, keep = [], None, None
    if rank == dst:
        arena = torch.empty(max(pos, 1), dtype=torch.uint8, device=dev)
        for r in range(world):
            if r != rank and H[r]:
                ops.append(dist.P2POp(dist.irecv, arena[base[r]: base[r] + H[r]], r, group=group))
    elif H[rank]:
        src = local
        if local.numel() < H[rank]:  # shorter than expected: the message keeps its hinted size, the tail is padding
            src = torch.zeros(H[rank], dtype=torch.uint8, device=dev)
            src[: local.numel()].copy_(local)
            keep = src
        ops.append(dist.P2POp(dist.isend, src[: H[rank]], dst, group=group))
    works = dist.batch_isend_irecv(ops) if ops else []

    def fin_hinted():
        length, status = read_meta(work_meta)
        pg.lengths = length
        totals = [sum(length[i] for i in owned[r]) for r in range(world)]
        for w in works:
            w.wait()
        if any(totals[r] > cap[r] for r in range(world)):  # somebody outgrew the arena: once more, exact lengths
            out, w2 = exact(length, status)
            for w in w2:
                w.wait()
            return out
        ops2 = []
        if rank == dst:
            offset = [0] * count
            for r in range(world):
                p = base[r]
                for i in owned[r]:
                    offset[i] = p
                    p += length[i]
                if r == rank:
                    arena[base[r]: base[r] + totals[r]].copy_(local)
                elif totals[r] > H[r]:
                    ops2.append(dist.P2POp(dist.irecv, arena[base[r] + H[r]: base[r] + totals[r]], r, group=group))
        elif totals[rank] > H[rank]:
            ops2.append(dist.P2POp(dist.isend, local[H[rank]:], dst, group=group))
        for w in (dist.batch_isend_irecv(ops2) if ops2 else []):
            w.wait()
        return (Gathered(arena, offset, length, status), keep)[0] if rank == dst else None  # (`keep`: the padded message lives until here)

    pg = PendingGather(fin_hinted)
    pg.lengths = None
    return pg


def gather_results(local, my_ids: Sequence[int], my_len: Sequence[int], my_status: Sequence[int], owned: List[List[int]],
                   count: int, group=None, dst: int = 0, async_op: bool = False, hint: Optional[Sequence[int]] = None):
    """The exchange step (see gather_begin).  Returns a Gathered on `dst`, None elsewhere; with async_op=True the
    PendingGather whose `finish()` returns that (bench.py overlaps the transfers with the next step's kernels and
    passes the step's lengths as the next step's `hint`, so no step waits for the size exchange)."""
    pg = gather_begin(local, my_ids, my_len, my_status, owned, count, group=group, dst=dst, hint=hint)
    return pg if async_op else pg.finish()


def run_sharded(buffers: Sequence[np.ndarray], engine: Callable[[np.ndarray], Tuple[int, object]], group=None,
                dst: int = 0, device=None) -> Optional[List[Tuple[int, np.ndarray]]]:
    """Runs `engine` (buffer -> (status, bytes)) on this rank's share and gathers everything on `dst`.

    Every rank passes the same `buffers` list (only the owned ones are touched).  `engine` may return its
    bytes as a numpy array or as a torch tensor on `device` (the GPU engines do: results stay in HBM until
    the destination asks for them).  Returns, on `dst`, a list of (status, bytes) in the original order;
    None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    count = len(buffers)
    owned = partition([len(b) for b in buffers], world)
    mine = owned[rank]
    dev = device if device is not None else torch.device("cpu")
    parts, lens, stats = [], [], []
    for i in mine:
        st, data = engine(buffers[i])
        t = data if isinstance(data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(data, dtype=np.uint8))
        parts.append(t.to(dev))
        lens.append(int(t.numel()))
        stats.append(int(st))
    local = torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=dev)
    g = gather_results(local, mine, lens, stats, owned, count, group=group, dst=dst)
    if g is None:
        return None
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    host = g.arena.cpu().numpy()
    return [(g.status[i], host[g.offset[i]: g.offset[i] + g.length[i]].copy()) for i in range(count)]


def gpu_engines(z, device):
    """(deflate_engine, inflate_engine) running one buffer through the HBM-resident C-ABI entry points;
    the results are CUDA tensors (they go into the gather without touching host memory)."""
    import torch

    def run_deflate(buf: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(buf)).to(device)
        try:
            return 0, z.deflate_tensor(t)
        except z.ZlibEsError as e:
            return e.code, torch.empty(0, dtype=torch.uint8, device=device)

    def run_inflate(buf: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(buf)).to(device)
        cap = max(4 * t.numel(), 1 << 20)
        for _ in range(8):
            out = torch.empty(cap, dtype=torch.uint8, device=device)
            try:
                return 0, z.inflate_tensor(t, out)
            except z.ZlibEsError as e:
                need = getattr(e, "need", 0)
                if e.code == z.ZES_E_NOSPACE and need > cap:
                    cap = need
                    continue
                return e.code, torch.empty(0, dtype=torch.uint8, device=device)
        return z.ZES_E_DEVICE, torch.empty(0, dtype=torch.uint8, device=device)

    return run_deflate, run_inflate


# ---------------------------------------------------------------------------------------------
# One buffer over several GPUs
# ---------------------------------------------------------------------------------------------
ADLER_MOD = 65521
BLOCK = 131072  # src/const.ts:7


def split_blocks(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous block ranges [first, last) per rank, as even as the block count allows."""
    nblk = (n + BLOCK - 1) // BLOCK
    per, extra = divmod(nblk, world)
    out, b = [], 0
    for r in range(world):
        k = per + (1 if r < extra else 0)
        out.append((b, b + k))
        b += k
    return out


def adler_combine(parts: Sequence[Tuple[int, int]]) -> int:
    """Adler-32 of a concatenation from the (adler32, length) of its pieces (src/adler32.ts:1-10 is
    associative in this sense: s1 adds up, s2 picks up len_after * (s1 - 1) of every piece before)."""
    s1, s2 = 1, 0
    for a, ln in parts:
        a1, a2 = a & 0xFFFF, (a >> 16) & 0xFFFF
        # appending a piece whose own sums started from s1 = 1: shift its s2 by len * (current s1 - 1)
        s2 = (s2 + a2 + (ln % ADLER_MOD) * ((s1 + ADLER_MOD - 1) % ADLER_MOD)) % ADLER_MOD
        s1 = (s1 + a1 + ADLER_MOD - 1) % ADLER_MOD
    return (s2 << 16) | s1


def join_host(pieces, bits: Sequence[int], adlers: Sequence[int], lens: Sequence[int]) -> np.ndarray:
    """The join on host memory (numpy): 78 9C | pieces bit-concatenated | zero pad | combined Adler-32 BE.
    The device form is zes_deflate_join_dev (z.deflate_join_tensors); this one serves callers without a GPU on the
    destination rank and the CPU test of the exchange."""
    total_bits = int(sum(bits))
    out = np.zeros(2 + (total_bits + 7) // 8 + 4, dtype=np.uint8)
    out[0], out[1] = 0x78, 0x9C  # src/zlib.ts:28-34
    body = out[2: len(out) - 4]
    pos = 0
    for p, nbits in zip(pieces, bits):
        if not nbits:
            continue
        p = np.asarray(p, dtype=np.uint8)
        nb = (nbits + 7) // 8
        sh, byte0 = pos & 7, pos >> 3
        w = p[:nb].astype(np.uint16) << sh  # the piece's bit k lands on bit pos + k: shift-merge at the seam
        body[byte0: byte0 + nb] |= (w & 0xFF).astype(np.uint8)
        hi8 = (w >> 8).astype(np.uint8)
        m = min(nb, len(body) - (byte0 + 1))
        body[byte0 + 1: byte0 + 1 + m] |= hi8[:m]
        assert not hi8[m:].any()
        pos += nbits
    a = adler_combine([(int(x), int(ln)) for x, ln in zip(adlers, lens) if ln])
    out[-4:] = np.frombuffer(int(a).to_bytes(4, "big"), dtype=np.uint8)  # src/zlib.ts:36-40
    return out


def deflate_split(n: int, deflate_range: Callable[[int, int, bool], Tuple[object, int, int]], join=join_host, group=None, dst: int = 0,
                  device=None):
    """deflate() of ONE n-byte buffer by all ranks of the group: bit-exact with the single-GPU result.

    Rank r takes the contiguous block range split_blocks(n, world)[r].  `deflate_range(lo, hi, final)` returns
    (bytes, nbits, adler32 of input[lo:hi]): the bit stream of the blocks of input[lo:hi] with BFINAL on the last
    block only if `final`, not padded beyond the last byte (z.deflate_range_tensor → zes_deflate_range_dev on a GPU;
    the oracle's zor_deflate_range in the gloo test).  The exchange: all_reduce of (nbits, adler32, length) per rank —
    the 8-element scan of SURVEY §8e — and the gather of the pieces (one message per rank, device to device under
    "nccl"); `dst` then calls `join(pieces, bits, adlers, lens)` — z.deflate_join_tensors on a GPU: every piece is
    shifted to its bit offset and the seam dwords are OR-ed (blocks are bit-concatenated: src/deflate.ts:20-37),
    `78 9C` in front, the combined Adler-32 behind (src/zlib.ts:28-46).  Returns join's result on `dst`, None
    elsewhere.  The reference's throw cases (n = 0, 1, n ≡ 1 mod 131072) are the caller's to reject first, exactly
    as zes_deflate does before it launches anything.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    ranges = split_blocks(n, world)
    b0, b1 = ranges[rank]
    lo, hi = b0 * BLOCK, min(n, b1 * BLOCK)
    last_rank = max(r for r in range(world) if ranges[r][1] > ranges[r][0])
    if b1 > b0:
        piece, nbits, ad = deflate_range(lo, hi, rank == last_rank)
        my = (int(nbits), int(ad), hi - lo)
    else:
        piece, my = np.zeros(0, dtype=np.uint8), (0, 1, 0)
    t = piece if isinstance(piece, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(piece, dtype=np.uint8))
    t = t.to(dev)
    pad = (-int(t.numel())) % 16  # pieces start on 16-byte boundaries of the destination's arena (the join reads dwords)
    if pad:
        t = torch.cat([t, torch.zeros(pad, dtype=torch.uint8, device=dev)])
    meta = torch.zeros(3 * world, dtype=torch.int64, device=dev)
    meta[3 * rank: 3 * rank + 3] = torch.tensor(my, dtype=torch.int64, device=dev)
    dist.all_reduce(meta, op=dist.ReduceOp.SUM, group=group)
    mh = [int(x) for x in meta.cpu().numpy()]
    g = gather_results(t, [rank], [int(t.numel())], [0], [[r] for r in range(world)], world, group=group, dst=dst)
    if g is None:
        return None
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    pieces = [g.result(r) for r in range(world)]
    if join is join_host:
        host = g.arena.cpu().numpy()
        pieces = [host[g.offset[r]: g.offset[r] + g.length[r]] for r in range(world)]
    return join(pieces, [mh[3 * r] for r in range(world)], [mh[3 * r + 1] for r in range(world)], [mh[3 * r + 2] for r in range(world)])


def split_bits(c: int, world: int) -> List[Tuple[int, int]]:
    """Ranges [lo_bit, own_bit) of a c-byte zlib stream per rank: equal byte shares, the first one behind the 2-byte
    header (src/zlib.ts:21).  A rank decodes the blocks that START inside its range."""
    cuts = [max(2, (c * r) // world) for r in range(world)] + [c]
    return [(8 * cuts[r], 8 * max(cuts[r], cuts[r + 1])) for r in range(world)]


def check_chain(meta: Sequence[Sequence[int]]) -> bool:
    """Do the per-rank results (ok, first_bit, end_bit, nblocks, out_len, final) make one stream?  The first block at
    bit 16, every range's end the next one's first, full 131072-byte blocks everywhere but at the very end, BFINAL
    on the last block and nowhere else (src/inflate.ts:22-37 stops at the first BFINAL)."""
    if not all(m[0] for m in meta):
        return False
    have = [m for m in meta if m[3] > 0]
    if not have or have[0][1] != 16:
        return False
    for a, b in zip(have, have[1:]):
        if a[2] != b[1] or a[5] or a[4] != a[3] * BLOCK:
            return False
    last = have[-1]
    return bool(last[5]) and (last[3] - 1) * BLOCK <= last[4] <= last[3] * BLOCK


def inflate_split(c: int, inflate_range: Callable[[int, int, bool], Optional[Tuple[object, int, int, int, int, bool]]],
                  fallback: Callable[[], object], group=None, dst: int = 0, device=None):
    """inflate() of ONE c-byte zlib stream by all ranks of the group: the same bytes as the
    single-GPU result.

    A reference-made stream is a chain of blocks of exactly 131072 output bytes (src/deflate.ts:20-37), so the blocks
    can be decoded anywhere once their first bits are known — and the engine's block-start search finds them in
    any part of the stream.  Rank r searches split_bits(c, world)[r] and decodes the blocks that start there:
    `inflate_range(lo_bit, own_bit, exact)` returns (bytes, out_len, first_bit, end_bit, nblocks, final) — bit
    positions relative to the stream — or None when its range is not a clean chain (z.inflate_range_tensor →
    zes_inflate_range_dev on a GPU).  The exchange: one all_reduce of those six numbers per rank; every rank then
    runs check_chain on the same table.  If the ranges fit, the outputs are gathered on `dst` in rank order (device
    to device under "nccl") and their concatenation is the result; if they do not (another encoder's stream, a rank
    that started on a false block start), `dst` returns `fallback()` — the single-GPU inflate of the whole stream,
    with the reference's error behaviour.  Returns the result on `dst`, None elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    lo, own = split_bits(c, world)[rank]
    res = inflate_range(lo, own, rank == 0) if own > lo else (np.zeros(0, dtype=np.uint8), 0, lo, lo, 0, False)
    if res is None:
        piece, my = np.zeros(0, dtype=np.uint8), (0, 0, 0, 0, 0, 0)
    else:
        piece, n, fb, eb, nb, fin = res
        my = (1, int(fb), int(eb), int(nb), int(n), int(bool(fin)))
    meta = torch.zeros(6 * world, dtype=torch.int64, device=dev)
    meta[6 * rank: 6 * rank + 6] = torch.tensor(my, dtype=torch.int64, device=dev)
    dist.all_reduce(meta, op=dist.ReduceOp.SUM, group=group)
    mh = [int(x) for x in meta.cpu().numpy()]
    table = [mh[6 * r: 6 * r + 6] for r in range(world)]
    if not check_chain(table):
        return fallback() if rank == dst else None
    t = piece if isinstance(piece, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(piece, dtype=np.uint8))
    t = t.to(dev)[: my[4]]
    g = gather_results(t, [rank], [int(t.numel())], [0], [[r] for r in range(world)], world, group=group, dst=dst)
    if g is None:
        return None
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return g.arena[: sum(g.length)]  # (the arena holds the ranks' outputs back to back, in rank order)
