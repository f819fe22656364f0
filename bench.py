#!/usr/bin/env python3
"""bench.py — deflate+inflate throughput of the MI355X DEFLATE engine on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic input: every rank deflates its
own HBM-resident 64 MiB buffer (BASELINE.json configs[1]: xorshift32 bytes, seed 12345+rank) and
inflates the result back, all through the C-ABI device entry points.  Buffers are independent
units: weak scaling, no data-path collective; the only exchange is an all-gather of
the per-shard compressed sizes (what a consumer needs to place the shards).

Rank 0 prints ONE JSON line.  `value` is GiB/s of uncompressed bytes taken through
deflate-then-inflate by the whole job (n * N * K / wall).  `roofline` prices the dominant kernel
(by HIP-event time on the library's stream) against HBM; `cpu_baseline` is the CPU oracle (a port
of the reference algorithm) timed on this box's host cores on the same buffer.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    # name: (generator, seed, bytes per buffer[, buffers per GPU])
    "random64": ("xorshift", 12345, 64 << 20),   # BASELINE.json configs[1] (the metric's config)
    "text64": ("itext", 12345, 64 << 20),        # configs[2]
    "lowent64": ("lowent4k", 12345, 64 << 20),
    "lowent256": ("lowent4k", 12345, 256 << 20),  # configs[4]: one of the 8 x 256 MiB buffers per GPU
    "batch1m": ("mix", 12345, 1 << 20, 128),      # configs[3]: this GPU's 128 of the 1024 x 1 MiB buffers, batch API
    # a stream another encoder made (CPython's zlib, level 6: history across blocks) — inflate only
    "zlibtext64": ("itext", 12345, 64 << 20, 1, "zlib6"),
}
MIX = ("xorshift", "itext", "lowent4k")           # buffer i uses seed 12345+i, generators in turn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="random64", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus %d needs torch.distributed.run (one process per GPU)" % args.gpus, file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    z = ge.load()
    z.init(local_rank)
    spec = WORKLOADS[args.workload]
    kind, seed, n1 = spec[0], spec[1], spec[2]
    nbuf = spec[3] if len(spec) > 3 else 1
    foreign = len(spec) > 4  # inflate-only workload: the compressed stream comes from CPython's zlib
    n = n1 * nbuf  # uncompressed bytes per GPU and step
    if nbuf == 1:
        host = z.gen(kind, seed + rank, n1)
    else:  # buffer i of the whole job: generator i % 3, seed + i
        host = np.concatenate([z.gen(MIX[(rank * nbuf + i) % 3], seed + rank * nbuf + i, n1) for i in range(nbuf)])
    d_in = torch.from_numpy(host).to(dev)
    d_foreign = None
    if foreign:
        import zlib as pyzlib

        foreign_bytes = pyzlib.compress(host.tobytes(), 6)
        d_foreign = torch.from_numpy(np.frombuffer(foreign_bytes, dtype=np.uint8).copy()).to(dev)
    bound1 = (z.deflate_bound(n1) + 15) // 16 * 16
    d_comp = torch.empty(bound1 * nbuf, dtype=torch.uint8, device=dev)
    d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    in_off = [i * n1 for i in range(nbuf)]
    c_off = [i * bound1 for i in range(nbuf)]
    state = {"clen": None}

    def run_deflate():
        """-> compressed bytes of this GPU's buffers (a view for one buffer, a count for a batch)"""
        if foreign:
            return d_foreign
        if nbuf == 1:
            return z.deflate_tensor(d_in, d_comp)
        clen, st = z.deflate_batch_tensor(d_in, in_off, [n1] * nbuf, d_comp, c_off, [bound1] * nbuf)
        assert not any(st), st
        state["clen"] = clen
        return clen

    def run_inflate(comp):
        if nbuf == 1:
            return z.inflate_tensor(comp, d_back)
        olen, st = z.inflate_batch_tensor(d_comp, c_off, comp, d_back, in_off, [n1] * nbuf)
        assert not any(st) and all(o == n1 for o in olen), (st[:4], olen[:4])
        return d_back

    def csize(comp):
        return int(comp.numel()) if nbuf == 1 else int(sum(comp))

    def step():
        comp = run_deflate()
        back = run_inflate(comp)
        if world > 1:  # exchange: every rank learns every shard's compressed size
            mine = torch.tensor([csize(comp)], dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(sizes, mine)
        return comp, back

    # --- verification (untimed): bit-exact vs the reference's own output, and round trip ---
    comp, back = step()
    c = csize(comp)
    verified = bool(back.numel() == n and bool((back == d_in).all()))
    golden_checked = False
    if rank == 0 and nbuf == 1 and not foreign:
        try:
            man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
            e = [x for x in man["big"] if x["kind"] == kind and x["seed"] == seed and x["n"] == n][0]
            digest = hashlib.sha256(comp.cpu().numpy().tobytes()).hexdigest()
            golden_checked = True
            verified = verified and c == e["deflate_len"] and digest == e["deflate_sha256"]
        except (OSError, IndexError, KeyError):
            pass

    for _ in range(args.warmup):
        step()

    z.set_profiling(True)
    ktimes = {}
    t_def = t_inf = 0.0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ta = time.perf_counter()
        comp = run_deflate()
        for name, ms, launches in ([] if foreign else z.last_kernel_times()):
            k = ktimes.setdefault(name, [0.0, 0])
            k[0] += ms
            k[1] += launches
        tb = time.perf_counter()
        back = run_inflate(comp)
        for name, ms, launches in z.last_kernel_times():
            k = ktimes.setdefault(name, [0.0, 0])
            k[0] += ms
            k[1] += launches
        tc = time.perf_counter()
        if world > 1:
            mine = torch.tensor([csize(comp)], dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(sizes, mine)
        t_def += tb - ta
        t_inf += tc - tb
    barrier()
    elapsed = time.perf_counter() - t0
    z.set_profiling(False)

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ok = torch.tensor([1 if verified else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        verified = bool(ok.item())

    if rank == 0:
        gib = float(1 << 30)
        value = n * world * args.steps / elapsed / gib
        # dominant kernel and its algorithmic traffic: each direction reads its input once and
        # writes its output once = (n + c) bytes per launch
        dom = max(ktimes.items(), key=lambda kv: kv[1][0]) if ktimes else None
        roofline = None
        if dom:
            name, (ms, launches) = dom
            avg_s = ms / 1e3 / max(launches, 1)
            achieved = (n + c) / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(args.workload, {}).get(name)
                except (OSError, ValueError):
                    traffic = None
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "algorithmic_bytes_per_launch": n + c, "avg_launch_ms": round(avg_s * 1e3, 4)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import _oracle  # CPU restatement of the reference algorithm: the checker, timed as the baseline

            ns = min(n, 64 << 20)  # bounded sample: at most 64 MiB of this GPU's input
            sample = host[:ns]
            t1 = time.perf_counter()
            oc = np.frombuffer(foreign_bytes, dtype=np.uint8) if foreign else _oracle.deflate(sample)
            t2 = time.perf_counter()
            ob = _oracle.inflate(oc)
            t3 = time.perf_counter()
            assert len(ob) == ns
            if foreign:
                cpu = {"value": round(ns / (t3 - t2) / gib, 5), "unit": "GiB/s", "cores": 1, "kind": "port",
                       "sample": "the whole %d MiB stream once: inflate %.2f s, 1 thread of %d host cores" % (ns >> 20, t3 - t2, os.cpu_count() or 0)}
            else:
                cpu = {"value": round(ns / (t3 - t1) / gib, 5), "unit": "GiB/s", "cores": 1, "kind": "port",
                       "sample": "the first %d MiB of the %s input once: deflate %.2f s + inflate %.2f s, 1 thread of %d host cores"
                                 % (ns >> 20, kind, t2 - t1, t3 - t2, os.cpu_count() or 0),
                       "deflate_gibs": round(ns / (t2 - t1) / gib, 5), "inflate_gibs": round(ns / (t3 - t2) / gib, 5)}
        # measured HBM copy bandwidth on this box (report against vendor peak and a measured copy)
        torch.cuda.synchronize()
        tcp = time.perf_counter()
        for _ in range(20):
            d_back.copy_(d_in)
        torch.cuda.synchronize()
        copy_gbs = 2 * n * 20 / (time.perf_counter() - tcp) / 1e9
        if roofline:
            roofline["measured_copy_GBs"] = round(copy_gbs, 1)
            roofline["frac_of_measured_copy"] = round(roofline["achieved"] / copy_gbs, 5)
        line = {
            "metric": ("GiB/s inflate of a 64 MiB zlib level-6 stream (uncompressed bytes / wall), output identical to the input" if foreign else
                       "GiB/s deflate+inflate round trip, %s (uncompressed bytes / wall), bit-exact vs reference"
                       % ("64 MiB buffers" if args.workload.endswith("64") else args.workload)),
            "value": round(value, 4),
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%s: %d x %d MiB %s buffer(s) per GPU (seed %d+index), %s, HBM-resident"
                                   % (args.workload, nbuf, n1 >> 20, kind, seed,
                                      "compressed by CPython zlib level 6, inflate only" if foreign else "deflate then inflate"), "buffers_per_step": world * nbuf, "bytes_per_buffer": n1,
                       "compressed_bytes": c, "parallelism": "independent buffers, one per GPU"},
            "deflate_gibs_per_gpu": None if foreign else round(n * args.steps / t_def / gib, 4),
            "inflate_gibs_per_gpu": round(n * args.steps / t_inf / gib, 4),
            "verified_bit_exact": verified,
            "golden_sha256_checked": golden_checked,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1][0])},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not verified:
        sys.exit(1)


if __name__ == "__main__":
    main()
