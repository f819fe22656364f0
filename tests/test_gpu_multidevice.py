"""One process, several GPUs (`zes_init(int ngpus)`, "the batch API is where multi-GPU concurrency lives"):
zes_init_devices gives the library one context per device, and the host batch forms partition their buffers over all of
them.  On a one-GPU box the contexts share the device (ZES_OVERSUBSCRIBE=1): the same partition, threads, per-context
streams, scratch pools and locks.  Each test runs in a process of its own (the device binding is per process)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PY = r'''
import os, sys, threading
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch
torch.cuda.init()
import __graft_entry__ as ge
import _oracle
z = ge.load()
ndev = z.init_devices(%(ndev)d)
assert ndev == %(ndev)d, ndev
specs = [("itext", 1, 300000), ("xorshift", 2, 131074), ("lowent4k", 3, 9000), ("itext", 4, 1), ("xorshift", 5, 300), ("itext", 6, 262144),
         ("lowent4k", 7, 2), ("xorshift", 8, 1 << 20), ("itext", 9, 1 << 20), ("lowent4k", 10, 700001), ("xorshift", 11, 131073), ("itext", 12, 65536)]
bufs = [z.gen(k, s, n) for k, s, n in specs]
def want(b):
    try:
        return _oracle.deflate(b).tobytes()
    except _oracle.OracleError as e:
        return e.code
res = z.deflate_batch(bufs)
def code(r):
    return r.code if isinstance(r, z.ZlibEsError) else 0
for b, r in zip(bufs, res):
    w = want(b)
    assert (code(r) if code(r) else r.tobytes()) == w, (len(b), code(r))
streams = [r for r in res if code(r) == 0]
plain = [b for b, r in zip(bufs, res) if code(r) == 0]
back = z.inflate_batch(streams)
for b, r in zip(plain, back):
    assert code(r) == 0 and r.tobytes() == b.tobytes()
# a damaged stream keeps its own error, whichever device it lands on
bad = [s.copy() for s in streams]
bad[2][0] = 0x77
bad[5] = bad[5][: len(bad[5]) // 2]
rb = z.inflate_batch(bad)
for k, r in enumerate(rb):
    try:
        w = (0, _oracle.inflate(bad[k]).tobytes())
    except _oracle.OracleError as e:
        w = (e.code, b"")
    assert (code(r), r.tobytes() if code(r) == 0 else b"") == w, k
# single host calls from several threads: the devices in turn
errs = []
def worker(i):
    try:
        for b in plain[i::4]:
            c = z.deflate(b)
            assert c.tobytes() == want(b)
            assert z.inflate(c).tobytes() == b.tobytes()
    except Exception as e:
        errs.append(repr(e))
ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not errs, errs
# device-pointer calls go to the device that holds the memory
t = torch.from_numpy(plain[0]).cuda()
c = z.deflate_tensor(t)
assert c.cpu().numpy().tobytes() == want(plain[0])
print("multi-device python checks passed")
'''


def _env(ndev):
    return dict(os.environ, ZES_OVERSUBSCRIBE="1", ZES_TEST_DEVICES=str(ndev))


@pytest.mark.parametrize("ndev", [2, 3])
def test_host_batches_over_several_contexts_match_the_oracle(gpu, ndev):
    out = subprocess.run([sys.executable, "-c", PY % {"root": ROOT, "ndev": ndev}], capture_output=True, text=True, timeout=600, env=_env(ndev))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "multi-device python checks passed" in out.stdout


def test_node_batch_uses_every_device(gpu, z, tmp_path):
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this box")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "zlib.es_amd", "host")])
    mix = ("xorshift", "itext", "lowent4k")
    path = str(tmp_path / "batch1m.bin")
    with open(path, "wb") as f:
        for i in range(32):
            f.write(z.gen(mix[i % 3], 12345 + i, 1 << 20).tobytes())
    env = dict(_env(3), ZES_BATCH1M_INPUT=path)
    out = subprocess.run([node, os.path.join(ROOT, "tests", "host_node_multidev_test.js")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "multi-device host checks passed" in out.stdout
