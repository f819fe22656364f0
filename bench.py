#!/usr/bin/env python3
"""bench.py — deflate+inflate throughput of the MI355X DEFLATE engine on BASELINE.json's workloads.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --inproc            # ONE process driving N GPUs through the host batch API (row n1)

One step = one pass of the hot path over one batch of synthetic input: every rank deflates its own
HBM-resident 64 MiB buffer (BASELINE.json configs[1]: xorshift32 bytes, seed 12345+rank) and inflates
the result back, all through the C-ABI device entry points.  Buffers are independent units (SURVEY
§8e): weak scaling, no data-path collective.  With N > 1 every step ends with the result gather of
the north_star: the compressed shards go from every rank's HBM to rank 0's HBM (zlib.es_amd/shard.py:
all_reduce of the sizes, one exact-length send per rank over xGMI), in flight while the next step's
kernels run and finished inside the timed region.

Rank 0 prints ONE JSON line.  `value` is GiB/s of uncompressed bytes taken through
deflate-then-inflate by the whole job (n * N * K / wall).  Every timed loop (exactly K steps between
barrier + synchronize) runs REPS = 3 times: the line carries the median, and the minimum and maximum
beside it, so that a slow box and a slow build can be told apart.  `roofline` prices the dominant
kernel (by HIP-event time on the library's stream) against HBM; `cpu_baseline` is the CPU oracle (a
port of the reference algorithm) timed on this box's host cores on a bounded sample of the same input.

The same process then runs the other BASELINE.json configurations as further legs of the same line,
each with its own golden flag, roofline and cpu_baseline:
  "text64"     configs[2]: one 64 MiB text-like buffer per GPU — deflate loop, then the inflate-only hot loop
  "batch1m"    configs[3]: this GPU's 128 of the 1024 x 1 MiB buffers through the batch entry points
  "lowent256"  configs[4]: one of the 8 x 256 MiB low-entropy buffers
  "zlibtext64" SURVEY §8f.1: a 64 MiB stream another encoder made (CPython zlib -6), inflate only
  "host_api"   (N = 1) the host-pointer path deflate(Uint8Array) / inflate(Uint8Array) actually is
               (reference src/zlib.ts:11,25): zes_deflate / zes_inflate_alloc on pageable and on pinned host
               arrays, PCIe included, random64 + text64 — and the same through Node when `node` is on the box.
`--workload X` makes X the only leg.
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
GIB = float(1 << 30)
REPS = 3               # every timed loop runs this often: median reported, min / max beside it

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
EXTRA_LEGS = ("text64", "batch1m", "lowent256", "zlibtext64")
LEG_TEXT = {
    "text64": "one 64 MiB itext buffer per GPU (seed 12345+rank), BASELINE.json configs[2]; deflate loop, then the inflate-only hot loop over the reference-format stream, HBM-resident",
    "batch1m": "this GPU's 128 of the 1024 x 1 MiB buffers of BASELINE.json configs[3] (buffer i: generator i % 3, seed 12345+i) through zes_deflate_batch_dev / zes_inflate_batch_dev, HBM-resident",
    "lowent256": "one 256 MiB lowent4k buffer per GPU (seed 12345+rank), BASELINE.json configs[4]; deflate loop, inflate loop, HBM-resident",
    "lowent64": "one 64 MiB lowent4k buffer per GPU; deflate loop, inflate loop, HBM-resident",
    "random64": "one 64 MiB xorshift32 buffer per GPU (seed 12345+rank), BASELINE.json configs[1]; deflate loop, inflate loop, HBM-resident",
    "zlibtext64": "64 MiB of itext compressed by CPython zlib level 6 (history across blocks, SURVEY §8f.1), inflate-only loop, HBM-resident",
}


def _median(xs):
    s = sorted(xs)
    return s[len(s) // 2]


def _spread(xs, scale=1.0, nd=4):
    """median / min / max of the repeats of one timed loop"""
    return {"median": round(_median(xs) * scale, nd), "min": round(min(xs) * scale, nd), "max": round(max(xs) * scale, nd)}


class Leg:
    """One workload on this rank: buffers in HBM, verification, the timed loops."""

    def __init__(self, name, env):
        import numpy as np
        import torch

        self.name, self.env = name, env
        z, dev, rank = env["z"], env["dev"], env["rank"]
        spec = WORKLOADS[name]
        self.kind, self.seed, self.n1 = spec[0], spec[1], spec[2]
        self.nbuf = spec[3] if len(spec) > 3 else 1
        self.foreign = len(spec) > 4  # inflate-only workload: the compressed stream comes from CPython's zlib
        self.n = self.n1 * self.nbuf  # uncompressed bytes per GPU and step
        if self.nbuf == 1:
            self.host = z.gen(self.kind, self.seed + rank, self.n1)
        else:  # buffer i of the whole job: generator i % 3, seed + i
            self.host = np.concatenate([z.gen(MIX[(rank * self.nbuf + i) % 3], self.seed + rank * self.nbuf + i, self.n1)
                                        for i in range(self.nbuf)])
        self.d_in = torch.from_numpy(self.host).to(dev)
        self.d_foreign = self.foreign_bytes = None
        if self.foreign:
            import zlib as pyzlib

            self.foreign_bytes = pyzlib.compress(self.host.tobytes(), 6)
            self.d_foreign = torch.from_numpy(np.frombuffer(self.foreign_bytes, dtype=np.uint8).copy()).to(dev)
        self.bound1 = (z.deflate_bound(self.n1) + 15) // 16 * 16
        # two result arenas: with N > 1 the gather of step k reads one while step k+1 writes the other
        self.d_comp = [torch.empty(self.bound1 * self.nbuf, dtype=torch.uint8, device=dev) for _ in range(2 if env["world"] > 1 else 1)]
        self.d_back = torch.empty(self.n, dtype=torch.uint8, device=dev)
        self.in_off = [i * self.n1 for i in range(self.nbuf)]
        self.c_off = [i * self.bound1 for i in range(self.nbuf)]
        self.ktimes = {"deflate": {}, "inflate": {}}

    # ---- the two directions ----
    def run_deflate(self, slot=0):
        """-> compressed bytes of this GPU's buffers (a view for one buffer, a list of lengths for a batch)"""
        z = self.env["z"]
        if self.foreign:
            return self.d_foreign
        if self.nbuf == 1:
            return z.deflate_tensor(self.d_in, self.d_comp[slot])
        clen, st = z.deflate_batch_tensor(self.d_in, self.in_off, [self.n1] * self.nbuf, self.d_comp[slot], self.c_off, [self.bound1] * self.nbuf)
        assert not any(st), st
        return clen

    def run_inflate(self, comp, slot=0):
        z = self.env["z"]
        if self.nbuf == 1:
            return z.inflate_tensor(comp, self.d_back)
        olen, st = z.inflate_batch_tensor(self.d_comp[slot], self.c_off, comp, self.d_back, self.in_off, [self.n1] * self.nbuf)
        assert not any(st) and all(o == self.n1 for o in olen), (st[:4], olen[:4])
        return self.d_back

    def csize(self, comp):
        return int(comp.numel()) if self.nbuf == 1 else int(sum(comp))

    def local_result(self, comp, slot):
        """This rank's compressed results back to back, as one device tensor (what the gather sends)."""
        import torch

        if self.nbuf == 1:
            return comp, [int(comp.numel())]
        arena = self.d_comp[slot]
        return torch.cat([arena[o:o + l] for o, l in zip(self.c_off, comp)]), [int(x) for x in comp]

    def note_times(self, direction):
        for name, ms, launches in self.env["z"].last_kernel_times():
            k = self.ktimes[direction].setdefault(name, [0.0, 0])
            k[0] += ms
            k[1] += launches

    # ---- verification (untimed): bit-exact vs the reference's own output, and round trip ----
    def verify(self):
        comp = self.run_deflate()
        back = self.run_inflate(comp)
        self.c = self.csize(comp)
        ok = bool(back.numel() == self.n and bool((back == self.d_in).all()))
        golden = False
        self.golden_note = None
        rank = self.env["rank"]
        gdir = os.path.join(ROOT, "tests", "golden")
        try:
            if self.foreign:  # the reference's own inflate of this very stream (foreign_big.json, make_zlibtext64.py)
                e = [x for x in json.load(open(os.path.join(gdir, "foreign_big.json")))
                     if x["kind"] == self.kind and x["seed"] == self.seed + rank and x["n"] == self.n][0]
                if hashlib.sha256(self.foreign_bytes).hexdigest() == e["stream_sha256"]:  # same zlib build, same stream
                    digest = hashlib.sha256(back.cpu().numpy().tobytes()).hexdigest()
                    ok = ok and back.numel() == e["output_len"] and digest == e["output_sha256"]
                    golden = True
                else:
                    self.golden_note = ("this box's CPython zlib made a different stream than the one the golden was made for (zlib 1.2.11): "
                                        "the output is checked against the input only (round trip)")
            elif self.nbuf == 1:  # every rank against the reference's output for ITS seed (rank 0: manifest.big)
                man = json.load(open(os.path.join(gdir, "manifest.json")))["big"]
                rk = os.path.join(gdir, "ranks_%s.json" % self.kind)
                if os.path.exists(rk):
                    man = man + json.load(open(rk))
                e = [x for x in man if x["kind"] == self.kind and x["seed"] == self.seed + rank and x["n"] == self.n][0]
                digest = hashlib.sha256(comp.cpu().numpy().tobytes()).hexdigest()
                ok = ok and self.c == e["deflate_len"] and digest == e["deflate_sha256"]
                golden = True
            else:  # this rank's share of configs[3]: every buffer against the reference's own output
                gold = json.load(open(os.path.join(gdir, "batch1m.json")))[rank * self.nbuf: (rank + 1) * self.nbuf]
                if len(gold) == self.nbuf:
                    hostc = self.d_comp[0].cpu().numpy()
                    for k, e in enumerate(gold):
                        assert e["i"] == rank * self.nbuf + k and e["seed"] == self.seed + e["i"]
                        digest = hashlib.sha256(hostc[self.c_off[k]: self.c_off[k] + comp[k]].tobytes()).hexdigest()
                        ok = ok and comp[k] == e["deflate_len"] and digest == e["deflate_sha256"]
                    golden = True
        except (OSError, IndexError, KeyError):
            pass
        self.verified, self.golden_checked = ok, golden
        return ok

    # ---- records ----
    def roofline(self, direction, c):
        """Dominant kernel of one direction (HIP events on the library's stream) against HBM: each direction reads its
        input once and writes its output once = (n + c) bytes per launch."""
        kt = self.ktimes[direction]
        if not kt:
            return None
        name, (ms, launches) = max(kt.items(), key=lambda kv: kv[1][0])
        avg_s = ms / 1e3 / max(launches, 1)
        achieved = (self.n + c) / avg_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(self.name, {}).get(name)
            except (OSError, ValueError):
                traffic = None
        return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": None if traffic is None else "profiles/traffic.json: rocprofv3 --pmc passes of this workload (profiles/collect.sh), not collected in this run",
                "algorithmic_bytes_per_launch": self.n + c, "avg_launch_ms": round(avg_s * 1e3, 4)}

    def cpu_baseline(self, cap=64 << 20):
        """The oracle (port of the reference algorithm, 1 thread) on a bounded sample of this leg's input."""
        import numpy as np

        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle  # CPU restatement of the reference algorithm: the checker, timed as the baseline

        ns = min(self.n, cap)  # bounded sample of this GPU's input
        sample = self.host[:ns]
        t1 = time.perf_counter()
        oc = np.frombuffer(self.foreign_bytes, dtype=np.uint8) if self.foreign else _oracle.deflate(sample)
        t2 = time.perf_counter()
        ob = _oracle.inflate(oc)
        t3 = time.perf_counter()
        assert len(ob) == (self.n if self.foreign else ns)
        cores = os.cpu_count() or 0
        if self.foreign:
            return {"value": round(self.n / (t3 - t2) / GIB, 5), "unit": "GiB/s", "cores": 1, "kind": "port",
                    "sample": "the whole %d MiB stream once: inflate %.2f s, 1 thread of %d host cores" % (self.n >> 20, t3 - t2, cores)}
        return {"value": round(ns / (t3 - t1) / GIB, 5), "unit": "GiB/s", "cores": 1, "kind": "port",
                "sample": "the first %d MiB of the %s input once: deflate %.2f s + inflate %.2f s, 1 thread of %d host cores"
                          % (ns >> 20, self.kind, t2 - t1, t3 - t2, cores),
                "deflate_gibs": round(ns / (t2 - t1) / GIB, 5), "inflate_gibs": round(ns / (t3 - t2) / GIB, 5)}


def _free_port():
    import socket

    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_command(ngpus, argv, port):
    """What `python bench.py --gpus N` (N > 1, not yet under a launcher) runs as its child: the driver's own form."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(args, argv, run=None):
    """Parent side of `--gpus N` without a launcher.  No GPU call is made in this process (no torch.cuda, no zes_init:
    a process that has initialised the GPU must not start the ranks by exec, and need not start them at all).
    The child's stdout is relayed; if the ranks fail with the gather on, they are started once more with
    --no-gather so that a transport problem costs the gather's measurement, not the scaling point — but the failure is
    not hidden: the relayed line carries `first_launch_failed` (the first launch's exit code) and the tail of its
    stderr, and the exit code of this process is then 3 whatever the second launch returned."""
    import subprocess
    import tempfile

    run = run or subprocess.run
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    tries = [list(argv)] + ([] if args.no_gather else [list(argv) + ["--no-gather"]])
    rc = 1
    first_rc, first_err = None, ""
    for k, av in enumerate(tries):
        cmd = launch_command(args.gpus, av, _free_port())
        print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        with tempfile.TemporaryFile(mode="w+") as errf:
            res = run(cmd, env=env, stdout=subprocess.PIPE, stderr=errf, text=True)
            errf.seek(0)
            err_text = errf.read()
        sys.stderr.write(err_text)
        rc = res.returncode
        lines = [ln for ln in (res.stdout or "").splitlines() if ln.startswith("{") and '"metric"' in ln]
        if lines:
            line = lines[-1]
            if k > 0:  # say so in the line itself, and in the exit code
                rec = json.loads(line)
                rec["first_launch_failed"] = first_rc
                rec["first_launch_stderr_tail"] = first_err[-1500:]
                rec["config"]["gather"] = "FAILED with the gather on (rc %d of the first launch); measured again with --no-gather" % first_rc
                line = json.dumps(rec)
                print(line, flush=True)
                return 3
            print(line, flush=True)
            return rc
        first_rc, first_err = rc, err_text
        print("bench.py: the ranks ended with rc %d and no result line%s" % (rc, "; once more without the gather" if k + 1 < len(tries) else ""),
              file=sys.stderr, flush=True)
    return rc or 1


# ------------------------------------------------------------------------------------------------------------------
# the host-pointer path (N = 1): what deflate(Uint8Array) / inflate(Uint8Array) of src/zlib.ts:11,25 are
# ------------------------------------------------------------------------------------------------------------------
def host_api_leg(z, dev, steps, hosts, goldens, cpu):
    """zes_deflate / zes_inflate_alloc on host arrays: pageable (a caller's ordinary array) and pinned (zes_host_alloc),
    the random64 and the text64 input, PCIe trips included; output arrays exist and have been touched before the clock
    starts (a binding that allocates a fresh 128 MiB array per call pays that array's page faults, not the library's
    time).  K calls per timed loop, REPS loops; then the same four rows through Node's deflate() / inflate() when `node`
    and the addon are on the box."""
    import ctypes as C

    import numpy as np
    import torch

    L = z.lib()
    rows = {}
    ok_all = True
    k = max(1, min(steps, 10))
    # the link itself on this box: 64 MiB of pinned memory up and down
    pin = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
    dbuf = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    up = down = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dbuf.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pin.copy_(dbuf, non_blocking=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        up, down = min(up, t1 - t0), min(down, t2 - t1)
    link = {"h2d_GBs": round((64 << 20) / up / 1e9, 1), "d2h_GBs": round((64 << 20) / down / 1e9, 1), "what": "64 MiB of pinned memory, torch copy, best of 4"}
    del pin, dbuf
    for wname, src in hosts.items():
        n = int(src.size)
        cap = z.deflate_bound(n)
        gold = goldens.get(wname)
        for pinned in (False, True):
            nb = n + n // 8 + (1 << 20)  # room for the early upper estimate of ZES_F_ALLOC_BOUND (the result is its first n bytes)
            if pinned:
                a, comp, back = z.host_alloc(n), z.host_alloc(cap), z.host_alloc(nb)
                a[:] = src
                comp[:] = 0
                back[:] = 0
            else:
                a, comp, back = src, np.zeros(cap, dtype=np.uint8), np.zeros(nb, dtype=np.uint8)
            clen, blen = C.c_uint64(), C.c_uint64()

            def alloc(_user, _index, need, back=back):  # the allocator callback of zes_inflate_alloc: the caller's (touched) array
                return back.ctypes.data if int(need) <= back.size else None

            cb = z.ALLOC_FN(alloc)
            assert L.zes_deflate(a.ctypes.data, n, comp.ctypes.data, cap, C.byref(clen)) == 0  # warm-up: pools, staging ring
            assert L.zes_inflate_alloc(comp.ctypes.data, clen.value, cb, None, C.byref(blen), 0) == 0 and blen.value == n
            td, ti, tx = [], [], []
            for _ in range(REPS):
                t0 = time.perf_counter()
                for _ in range(k):
                    rc = L.zes_deflate(a.ctypes.data, n, comp.ctypes.data, cap, C.byref(clen))
                t1 = time.perf_counter()
                assert rc == 0
                for _ in range(k):  # the allocator may be asked early for an upper estimate: what the N-API addon does
                    rc = L.zes_inflate_alloc(comp.ctypes.data, clen.value, cb, None, C.byref(blen), z.ZES_F_ALLOC_BOUND)
                t2 = time.perf_counter()
                assert rc == 0 and blen.value == n
                for _ in range(k):  # the allocator is asked once, for the exact size (known last: the download cannot overlap the decode)
                    rc = L.zes_inflate_alloc(comp.ctypes.data, clen.value, cb, None, C.byref(blen), 0)
                t3 = time.perf_counter()
                assert rc == 0 and blen.value == n
                td.append((t1 - t0) / k)
                ti.append((t2 - t1) / k)
                tx.append((t3 - t2) / k)
            c = int(clen.value)
            ok = bool((back[:n] == src).all())
            gchk = False
            if gold:
                ok = ok and c == gold["deflate_len"] and hashlib.sha256(comp[:c].tobytes()).hexdigest() == gold["deflate_sha256"]
                gchk = True
            ok_all = ok_all and ok
            md, mi = _median(td), _median(ti)
            rows["%s_%s" % (wname, "pinned" if pinned else "pageable")] = {
                "deflate_gibs": round(n / md / GIB, 3), "inflate_gibs": round(n / mi / GIB, 3),
                "inflate_exact_alloc_gibs": round(n / _median(tx) / GIB, 3),
                "deflate_ms": _spread(td, 1e3), "inflate_ms": _spread(ti, 1e3), "inflate_exact_alloc_ms": _spread(tx, 1e3), "compressed_bytes": c,
                "verified_bit_exact": ok, "golden_sha256_checked": gchk,
                # bytes over the link per call: n up + c down (deflate), c up + n down (inflate); a full-duplex link's
                # floor is max(up, down) per direction
                "roofline": {"bound": "pcie", "unit": "GB/s", "peak": link["h2d_GBs"],
                             "achieved_deflate": round((n + c) / md / 1e9, 2), "achieved_inflate": round((n + c) / mi / 1e9, 2),
                             "frac_deflate": round(n / md / 1e9 / link["h2d_GBs"], 4), "frac_inflate": round(n / mi / 1e9 / link["d2h_GBs"], 4),
                             "note": "frac = the larger of the two transfers of the call (n bytes) per second against the measured one-way link rate"},
            }
            if pinned:
                for x in (a, comp, back):
                    z.host_free(x)
    out = {"workload": "host-pointer API, 64 MiB calls, PCIe included (never `value`): zes_deflate / zes_inflate_alloc (ZES_F_ALLOC_BOUND, and the exact-size protocol beside it) on pageable and pinned host arrays, "
                       "%d calls per timed loop, median of %d loops; reference src/zlib.ts:11,25" % (k, REPS),
           "link": link, "rows": rows, "verified_bit_exact": ok_all,
           "cpu_baseline": None if not cpu else {w: cpu.get(w) for w in hosts}}
    out["node"] = node_host_leg(hosts, goldens, k)
    if out["node"] and out["node"].get("verified_bit_exact") is False:
        out["verified_bit_exact"] = False
    return out


def node_host_leg(hosts, goldens, k):
    """deflate(Uint8Array) / inflate(Uint8Array) of the TypeScript façade under Node (zlib.es_amd/host/zlib.js over the
    N-API addon): the same inputs from files, timed inside Node (zlib.es_amd/host/bench_host.js).  A child process."""
    import shutil
    import subprocess
    import tempfile

    node = shutil.which("node")
    addon = os.path.join(ROOT, "zlib.es_amd", "host", "build", "zes_napi.node")
    script = os.path.join(ROOT, "zlib.es_amd", "host", "bench_host.js")
    if node is None or not os.path.exists(addon) or not os.path.exists(script):
        return {"skipped": "node or the addon is not on this box"}
    tmp = tempfile.mkdtemp(prefix="zes_bench_")
    try:
        spec = []
        for w, src in hosts.items():
            path = os.path.join(tmp, w + ".bin")
            src.tofile(path)
            g = goldens.get(w) or {}
            spec.append({"name": w, "path": path, "deflate_len": g.get("deflate_len"), "deflate_sha256": g.get("deflate_sha256")})
        res = subprocess.run([node, script, json.dumps({"inputs": spec, "calls": k, "reps": REPS})], capture_output=True, text=True, timeout=240)
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        if res.returncode != 0 or not lines:
            return {"failed": "rc %d: %s" % (res.returncode, (res.stderr or res.stdout)[-400:]), "verified_bit_exact": False}
        return json.loads(lines[-1])
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return {"failed": repr(e)[:300], "verified_bit_exact": False}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------------------------------------
# one process, N GPUs (row n1: "host code stays TypeScript on Node ... shard across the 8 GPUs"): no RCCL on this road
# ------------------------------------------------------------------------------------------------------------------
def inproc_plan(ngpus, nbuf_per_gpu=128, n1=1 << 20):
    """(generator, seed) of every buffer and the owner the library's partition rule gives it: BASELINE configs[3]'s
    buffers, 128 per GPU (all 1024 at N = 8)."""
    total = ngpus * nbuf_per_gpu
    return [(MIX[i % 3], 12345 + i, n1) for i in range(total)]


def main_inproc(args):
    """`--gpus N --inproc`: ONE process, zes_init_devices(N), the configs[3] batch (128 x 1 MiB per GPU) through the host
    batch forms zes_deflate_batch / zes_inflate_batch_alloc from pinned host arrays — the path a Node host takes — every
    buffer against the reference-run golden.  Same JSON shape as the torchrun line."""
    import ctypes as C

    import numpy as np

    import __graft_entry__ as ge

    z = ge.load()
    L = z.lib()
    got = z.init_devices(args.gpus)
    plan = inproc_plan(args.gpus)
    cnt = len(plan)
    n1 = plan[0][2]
    cap1 = (z.deflate_bound(n1) + 15) // 16 * 16
    src = z.host_alloc(cnt * n1)
    comp = z.host_alloc(cnt * cap1)
    back = z.host_alloc(cnt * n1)
    for i, (kind, seed, n) in enumerate(plan):
        src[i * n1:(i + 1) * n1] = z.gen(kind, seed, n)
    comp[:] = 0
    back[:] = 0
    P = C.c_void_p * cnt
    U = C.c_uint64 * cnt
    in_ptr = P(*[src.ctypes.data + i * n1 for i in range(cnt)])
    c_ptr = P(*[comp.ctypes.data + i * cap1 for i in range(cnt)])
    lens = U(*([n1] * cnt))
    caps = U(*([cap1] * cnt))
    clen, olen = U(), U()
    st = (C.c_int32 * cnt)()

    def alloc(_user, index, need):
        return back.ctypes.data + int(index) * n1 if int(need) <= n1 else None

    cb = z.ALLOC_FN(alloc)

    def deflate():
        rc = L.zes_deflate_batch(in_ptr, lens, c_ptr, caps, clen, st, cnt)
        assert rc == 0 and not any(st), (rc, list(st)[:8])

    def inflate():
        rc = L.zes_inflate_batch_alloc(c_ptr, clen, cb, None, olen, st, cnt, 0)
        assert rc == 0 and not any(st) and all(o == n1 for o in olen), (rc, list(st)[:8])

    deflate()
    inflate()
    ok = bool((back == src).all())
    golden = False
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "batch1m.json")))[:cnt]
        if len(gold) == cnt:
            for i, e in enumerate(gold):
                c = int(clen[i])
                ok = ok and e["i"] == i and c == e["deflate_len"] and hashlib.sha256(comp[i * cap1:i * cap1 + c].tobytes()).hexdigest() == e["deflate_sha256"]
            golden = True
    except (OSError, KeyError, ValueError):
        pass
    owners = z.partition([n1] * cnt, got)
    for _ in range(args.warmup):
        deflate()
        inflate()
    td, ti = [], []
    for _ in range(REPS):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            deflate()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            inflate()
        t2 = time.perf_counter()
        td.append(t1 - t0)
        ti.append(t2 - t1)
    total = cnt * n1
    md, mi = _median(td), _median(ti)
    ctot = int(sum(clen))
    line = {
        "metric": "GiB/s deflate+inflate round trip, 1 MiB buffers through the host batch API of ONE process driving N GPUs (uncompressed bytes / wall, PCIe included), bit-exact vs reference",
        "value": round(total * args.steps / (md + mi) / GIB, 4), "unit": "GiB/s", "n_gpus": got, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round((md + mi) / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "inproc batch1m: %d x 1 MiB buffers (BASELINE.json configs[3]: generator i %% 3, seed 12345+i; 128 per GPU) from pinned host arrays through "
                               "zes_deflate_batch / zes_inflate_batch_alloc, one process, zes_init_devices(%d)" % (cnt, got),
                   "buffers_per_step": cnt, "bytes_per_buffer": n1, "compressed_bytes": ctot,
                   "parallelism": "one host process, %d device contexts, buffers partitioned by size (zes_partition), no collective" % got,
                   "buffers_per_context": [owners.count(d) for d in range(got)],
                   "oversubscribed": bool(os.environ.get("ZES_OVERSUBSCRIBE")), "gather": None},
        "repeats": REPS,
        "deflate_gibs": round(total * args.steps / md / GIB, 4), "inflate_gibs": round(total * args.steps / mi / GIB, 4),
        "deflate_ms": _spread([t / args.steps for t in td], 1e3), "inflate_ms": _spread([t / args.steps for t in ti], 1e3),
        "verified_bit_exact": ok, "golden_sha256_checked": golden,
        "roofline": None, "cpu_baseline": None, "pool_bytes": z.pool_bytes(),
    }
    print(json.dumps(line), flush=True)
    for x in (src, comp, back):
        z.host_free(x)
    if not ok:
        sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the compressed shards where they are")
    ap.add_argument("--no-extra-legs", action="store_true", help="only the main leg and text64")
    ap.add_argument("--inproc", action="store_true", help="one process drives all N GPUs through the host batch API (no RCCL)")
    args = ap.parse_args()

    if args.inproc:
        return main_inproc(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process never touches the GPU — it starts the N ranks as a CHILD
        # (torch.distributed.run, one process per GPU) and relays rank 0's JSON line and the exit code.
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # ZES_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (the ranks share the devices,
    # the gather's messages go through host memory) — every line of the rank logic but RCCL itself; never a measurement
    backend = os.environ.get("ZES_BENCH_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the small reduction tensors live
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime

        # (a collective that hangs ends the run after three minutes instead of the default ten)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))

    z = ge.load()
    z.init(local_dev)
    env = {"z": z, "dev": dev, "rank": rank, "world": world}
    main_name = args.workload or "random64"
    extra = [] if args.workload else (["text64"] if args.no_extra_legs else list(EXTRA_LEGS))
    shard = None
    if world > 1 and not args.no_gather:
        import importlib.util

        sp = importlib.util.spec_from_file_location("zlibes_amd.shard", os.path.join(ROOT, "zlib.es_amd", "shard.py"))
        shard = importlib.util.module_from_spec(sp)
        sp.loader.exec_module(shard)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_time(elapsed, ok):
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
            okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = bool(okt.item())
        return elapsed, ok

    def reduce_flag(flag):
        """True only if true on every rank (e.g. "this rank's output matched the reference's for its own seed")."""
        if world > 1:
            t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            flag = bool(t.item())
        return bool(flag)

    def minmax_over_ranks(x):
        if world > 1:
            lo = torch.tensor([x], dtype=torch.float64, device=cdev)
            hi = lo.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            return float(lo.item()), float(hi.item())
        return x, x

    # ------------------------------------------------------------------ main leg
    leg = Leg(main_name, env)
    verified = leg.verify()
    leg.golden_checked = reduce_flag(leg.golden_checked)
    owned = [[r * leg.nbuf + i for i in range(leg.nbuf)] for r in range(world)]
    gathered_bytes = [0]

    hint = [None]  # the step before's lengths: the next gather posts its transfers from them without waiting for the sizes

    def finish(pending):
        if pending is None:
            return
        got = pending.finish()
        hint[0] = pending.lengths
        torch.cuda.current_stream(dev).synchronize()  # the library runs on its own stream: the host must know the transfer is over
        if got is not None:
            gathered_bytes[0] += int(sum(got.length))

    def step(k, pending, timed):
        slot = k & 1 if world > 1 else 0
        ta = time.perf_counter()
        comp = leg.run_deflate(slot)
        if timed and not leg.foreign:
            leg.note_times("deflate")
        tb = time.perf_counter()
        leg.run_inflate(comp, slot)
        if timed:
            leg.note_times("inflate")
        tc = time.perf_counter()
        if shard is not None:
            # the gather of the step before has had this step's kernels to hide behind; its arena is free again
            finish(pending)
            local, lens = leg.local_result(comp, slot)
            if backend != "nccl":
                local = local.cpu()
            pending = shard.gather_results(local, owned[rank], lens, [0] * leg.nbuf, owned, world * leg.nbuf, dst=0, async_op=True, hint=hint[0])
        return pending, tb - ta, tc - tb

    pending = None
    for k in range(args.warmup):
        pending, _, _ = step(k, pending, False)
    finish(pending)
    pending = None

    z.set_profiling(True)
    reps = []  # (elapsed, deflate seconds, inflate seconds) of each repeat of the K timed steps
    for _ in range(REPS):
        gathered_bytes[0] = 0
        t_def = t_inf = 0.0
        pending = None
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            pending, d, i = step(k, pending, True)
            t_def += d
            t_inf += i
        finish(pending)
        barrier()
        elapsed = time.perf_counter() - t0
        elapsed, verified = reduce_time(elapsed, verified)
        reps.append((elapsed, t_def, t_inf))
    z.set_profiling(False)
    elapsed, t_def, t_inf = sorted(reps)[len(reps) // 2]
    n_timed = REPS * args.steps  # launches the kernel-time sums cover
    main_pool = z.pool_bytes()
    d_lo, d_hi = (0.0, 0.0) if leg.foreign else minmax_over_ranks(leg.n * args.steps / t_def / GIB)
    i_lo, i_hi = minmax_over_ranks(leg.n * args.steps / t_inf / GIB)

    # ------------------------------------------------------------------ the other legs: deflate loop, inflate loop
    def run_extra(name):
        nonlocal verified
        tl = Leg(name, env)
        tok = tl.verify()
        tl.golden_checked = reduce_flag(tl.golden_checked)
        comp = tl.run_deflate()
        for _ in range(args.warmup):
            tl.run_deflate()
            tl.run_inflate(comp)
        z.set_profiling(True)
        tds, tis = [], []
        for _ in range(REPS):
            if not tl.foreign:
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    tl.run_deflate()
                    tl.note_times("deflate")
                barrier()
                td, tok = reduce_time(time.perf_counter() - t1, tok)
                tds.append(td)
            barrier()
            t2 = time.perf_counter()
            for _ in range(args.steps):  # the inflate-only hot loop
                tl.run_inflate(comp)
                tl.note_times("inflate")
            barrier()
            ti, tok = reduce_time(time.perf_counter() - t2, tok)
            tis.append(ti)
        z.set_profiling(False)
        verified = verified and tok
        rec = None
        if rank == 0:
            ti = _median(tis)
            rec = {
                "workload": "%s: %s" % (name, LEG_TEXT[name]),
                "inflate_gibs": round(tl.n * world * args.steps / ti / GIB, 4),
                "inflate_gibs_per_gpu": round(tl.n * args.steps / ti / GIB, 4),
                "inflate_ms": round(ti / args.steps * 1e3, 4),
                "inflate_ms_spread": _spread([t / args.steps for t in tis], 1e3),
            }
            if tds:
                td = _median(tds)
                rec.update({
                    "deflate_gibs": round(tl.n * world * args.steps / td / GIB, 4),
                    "deflate_gibs_per_gpu": round(tl.n * args.steps / td / GIB, 4),
                    "deflate_ms": round(td / args.steps * 1e3, 4),
                    "deflate_ms_spread": _spread([t / args.steps for t in tds], 1e3),
                    "round_trip_gibs_per_gpu": round(tl.n * args.steps / (td + ti) / GIB, 4),
                })
            rec.update({
                "repeats": REPS,
                "compressed_bytes": tl.c,
                "verified_bit_exact": tok,
                "golden_sha256_checked": tl.golden_checked,
                "roofline": tl.roofline("inflate", tl.c),
                "roofline_deflate": tl.roofline("deflate", tl.c),
                "cpu_baseline": None if (world > 1 or args.no_cpu_baseline) else tl.cpu_baseline(),
                "pool_bytes": z.pool_bytes(),
                "kernels_ms_per_step": {k: round(v[0] / (REPS * args.steps), 4) for d in ("deflate", "inflate")
                                        for k, v in sorted(tl.ktimes[d].items(), key=lambda kv: -kv[1][0])},
            })
            if tl.golden_note:
                rec["golden_note"] = tl.golden_note
        host = tl.host if name == "text64" else None
        del tl
        torch.cuda.empty_cache()
        z.trim()  # the next leg's pooled bytes are its own
        return rec, host

    legs = {}
    text_host = None
    z.trim()
    for name in extra:
        legs[name], h = run_extra(name)
        if h is not None:
            text_host = h

    if rank == 0:
        n, c = leg.n, leg.c
        value = n * world * args.steps / elapsed / GIB
        vals = [n * world * args.steps / r[0] / GIB for r in reps]
        both = dict(leg.ktimes["deflate"])
        both.update(leg.ktimes["inflate"])
        dom_dir = "inflate"
        if both:
            dom = max(both.items(), key=lambda kv: kv[1][0])[0]
            dom_dir = "deflate" if dom in leg.ktimes["deflate"] else "inflate"
        roofline = leg.roofline(dom_dir, c)
        cpu = None if (world > 1 or args.no_cpu_baseline) else leg.cpu_baseline()
        # measured HBM copy bandwidth on this box (report against vendor peak and a measured copy)
        torch.cuda.synchronize()
        tcp = time.perf_counter()
        for _ in range(20):
            leg.d_back.copy_(leg.d_in)
        torch.cuda.synchronize()
        copy_gbs = 2 * n * 20 / (time.perf_counter() - tcp) / 1e9
        for r in [roofline] + [x for lg in legs.values() if lg for x in (lg["roofline"], lg["roofline_deflate"])]:
            if r:
                r["measured_copy_GBs"] = round(copy_gbs, 1)
                r["frac_of_measured_copy"] = round(r["achieved"] / copy_gbs, 5)
        foreign = leg.foreign
        line = {
            "metric": ("GiB/s inflate of a 64 MiB zlib level-6 stream (uncompressed bytes / wall), output identical to the input" if foreign else
                       "GiB/s deflate+inflate round trip, %s (uncompressed bytes / wall), bit-exact vs reference"
                       % ("64 MiB buffers" if main_name.endswith("64") else main_name)),
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
                                   % (main_name, leg.nbuf, leg.n1 >> 20, leg.kind, leg.seed,
                                      "compressed by CPython zlib level 6, inflate only" if foreign else "deflate then inflate"),
                       "buffers_per_step": world * leg.nbuf, "bytes_per_buffer": leg.n1,
                       "compressed_bytes": c, "parallelism": "independent buffers, one per GPU",
                       **({} if backend == "nccl" else {"rehearsal": "ranks over %s, sharing the visible GPUs: not a measurement" % backend}),
                       "gather": (None if shard is None else
                                  "every step's compressed shards gathered into rank 0's HBM over RCCL/xGMI (one message per rank posted from the step "
                                  "before's lengths, sizes all_reduce read behind it), overlapped with the next step, completed inside the timed "
                                  "region: %d result bytes on rank 0 per step"
                                  % (gathered_bytes[0] // max(args.steps, 1)))},
            # every timed loop ran REPS times (exactly `steps` steps each): value / ms_per_step are the median repeat's
            "repeats": REPS,
            "value_spread": _spread(vals),
            "ms_per_step_spread": _spread([r[0] / args.steps for r in reps], 1e3),
            "deflate_gibs_per_gpu": None if foreign else round(n * args.steps / t_def / GIB, 4),
            "inflate_gibs_per_gpu": round(n * args.steps / t_inf / GIB, 4),
            "deflate_gibs_per_gpu_spread": None if foreign else _spread([n * args.steps / r[1] / GIB for r in reps]),
            "inflate_gibs_per_gpu_spread": _spread([n * args.steps / r[2] / GIB for r in reps]),
            # what a reader of an N > 1 line has to check: the process group's size, the devices the ranks saw, every rank's own rate
            "ranks": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": backend if world > 1 else None,
                      "device_count": torch.cuda.device_count(), "device": torch.cuda.get_device_name(dev),
                      "deflate_gibs_per_rank_min_max": None if foreign else [round(d_lo, 3), round(d_hi, 3)],
                      "inflate_gibs_per_rank_min_max": [round(i_lo, 3), round(i_hi, 3)],
                      "gathered_bytes_per_step": None if shard is None else gathered_bytes[0] // max(args.steps, 1)},
            "verified_bit_exact": verified,
            "golden_sha256_checked": leg.golden_checked,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "pool_bytes": main_pool,
            "kernels_ms_per_step": {k: round(v[0] / n_timed, 4) for k, v in sorted(both.items(), key=lambda kv: -kv[1][0])},
        }
        for name in extra:
            line[name] = legs[name]
        if world == 1 and extra and not args.no_extra_legs:
            # the host-pointer path: the same random64 / text64 inputs, PCIe included
            man = {}
            try:
                for e in json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["big"]:
                    if e["n"] == 64 << 20 and e["seed"] == 12345:
                        man[e["kind"]] = e
            except (OSError, KeyError, ValueError):
                pass
            hosts = {"random64": leg.host}
            if text_host is not None:
                hosts["text64"] = text_host
            cpus = {"random64": cpu, "text64": legs.get("text64") and legs["text64"]["cpu_baseline"]}
            try:
                line["host_api"] = host_api_leg(z, dev, args.steps, hosts, {"random64": man.get("xorshift"), "text64": man.get("itext")},
                                                None if args.no_cpu_baseline else cpus)
                if line["host_api"]["verified_bit_exact"] is False:
                    verified = False
                    line["verified_bit_exact"] = False
            except AssertionError as e:  # a failed host call fails the run, and says where
                line["host_api"] = {"failed": repr(e)[:300]}
                verified = False
                line["verified_bit_exact"] = False
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not verified:
        sys.exit(1)


if __name__ == "__main__":
    main()
