"""Stress of the several-contexts host paths (run on the GPU box; ZES_OVERSUBSCRIBE=1 on a one-GPU box):
python tools/gpu_multidev_stress.py [seconds] [contexts] — several Python threads issue single calls and batches at the same
time (sizes from a few bytes to 40 MiB: direct copies, the pinned ring, the runtime's copy path, the pipelined calls),
every result against the oracle (small) or a round trip (large)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import __graft_entry__ as ge
import _oracle
z = ge.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 3
assert z.init_devices(nctx) == nctx
kinds = ("itext", "lowent4k", "xorshift")
t_end = time.time() + budget
errs, counts = [], [0] * 4

def code(r):
    return r.code if isinstance(r, z.ZlibEsError) else 0

def worker(w):
    rng = np.random.default_rng(100 + w)
    try:
        while time.time() < t_end and not errs:
            if rng.integers(3) == 0:  # a batch
                bufs = []
                for _ in range(int(rng.integers(2, 20))):
                    n = int(rng.choice([2, 3, 1000, 65536, 131072, 131074, 300000, 1 << 20, int(rng.integers(2, 3 << 20))]))
                    if n % 131072 == 1:
                        n += 1
                    bufs.append(z.gen(kinds[int(rng.integers(3))], int(rng.integers(1 << 30)), n))
                res = z.deflate_batch(bufs)
                for b, r in zip(bufs, res):
                    assert code(r) == 0
                    if len(b) <= 400000:
                        assert r.tobytes() == _oracle.deflate(b).tobytes(), ("batch deflate", len(b))
                back = z.inflate_batch(res)
                for b, r in zip(bufs, back):
                    assert code(r) == 0 and r.tobytes() == b.tobytes(), ("batch inflate", len(b))
            else:  # a single call, now and then a long one (the pipelined paths: side threads, copy streams of the context)
                n = int(rng.choice([5, 70000, 1 << 20, 5 << 20, 9 << 20, int(rng.integers(2, 2 << 20)), (40 << 20) + 5 if rng.integers(6) == 0 else 300001]))
                a = z.gen(kinds[int(rng.integers(3))], int(rng.integers(1 << 30)), n)
                c = z.deflate(a)
                if n <= 400000:
                    assert c.tobytes() == _oracle.deflate(a).tobytes(), ("deflate", n)
                assert z.inflate(c).tobytes() == a.tobytes(), ("inflate", n)
            counts[w] += 1
    except Exception as e:  # noqa: BLE001
        errs.append("worker %d: %r" % (w, e))

ts = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
[t.start() for t in ts]
while any(t.is_alive() for t in ts):
    time.sleep(20)
    print("rounds", counts, flush=True)
[t.join() for t in ts]
print("multi-device stress:", "FAILED " + "; ".join(errs) if errs else "ok", counts, flush=True)
sys.exit(1 if errs else 0)
