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
    # name: (generator, seed, bytes, golden key)
    "random64": ("xorshift", 12345, 64 << 20),   # BASELINE.json configs[1] (the metric's config)
    "text64": ("itext", 12345, 64 << 20),        # configs[2]
    "lowent64": ("lowent4k", 12345, 64 << 20),
}


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
    kind, seed, n = WORKLOADS[args.workload]
    host = z.gen(kind, seed + rank, n)
    d_in = torch.from_numpy(host).to(dev)
    d_comp = torch.empty(z.deflate_bound(n), dtype=torch.uint8, device=dev)
    d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)

    def step():
        comp = z.deflate_tensor(d_in, d_comp)
        back = z.inflate_tensor(comp, d_back)
        if world > 1:  # exchange: every rank learns every shard's compressed size
            mine = torch.tensor([comp.numel()], dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(sizes, mine)
        return comp, back

    # --- verification (untimed): bit-exact vs the reference's own output, and round trip ---
    comp, back = step()
    c = int(comp.numel())
    verified = bool(back.numel() == n and bool((back == d_in).all()))
    golden_checked = False
    if rank == 0:
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
        comp = z.deflate_tensor(d_in, d_comp)
        for name, ms, launches in z.last_kernel_times():
            k = ktimes.setdefault(name, [0.0, 0])
            k[0] += ms
            k[1] += launches
        tb = time.perf_counter()
        back = z.inflate_tensor(comp, d_back)
        for name, ms, launches in z.last_kernel_times():
            k = ktimes.setdefault(name, [0.0, 0])
            k[0] += ms
            k[1] += launches
        tc = time.perf_counter()
        if world > 1:
            mine = torch.tensor([comp.numel()], dtype=torch.int64, device=dev)
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

            t1 = time.perf_counter()
            oc = _oracle.deflate(host)
            t2 = time.perf_counter()
            ob = _oracle.inflate(oc)
            t3 = time.perf_counter()
            assert len(ob) == n
            cpu = {"value": round(n / (t3 - t1) / gib, 5), "unit": "GiB/s", "cores": 1, "kind": "port",
                   "sample": "the full %d MiB %s buffer once: deflate %.2f s + inflate %.2f s, 1 thread of %d host cores"
                             % (n >> 20, kind, t2 - t1, t3 - t2, os.cpu_count() or 0),
                   "deflate_gibs": round(n / (t2 - t1) / gib, 5), "inflate_gibs": round(n / (t3 - t2) / gib, 5)}
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
            "metric": "GiB/s deflate+inflate round trip, 64 MiB buffers (uncompressed bytes / wall), bit-exact vs reference",
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
            "config": {"workload": "%s: one %d MiB %s buffer per GPU (seed %d+rank), deflate then inflate, HBM-resident"
                                   % (args.workload, n >> 20, kind, seed), "buffers_per_step": world, "bytes_per_buffer": n,
                       "compressed_bytes": c, "parallelism": "independent buffers, one per GPU"},
            "deflate_gibs_per_gpu": round(n * args.steps / t_def / gib, 4),
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
