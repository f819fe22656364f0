"""Which block starts of another encoder's stream does the block-start search find?  (run on the GPU box)
Compares the candidate lists of the segment-parallel tier (ZES_T2_DBG=1) with the oracle's map of the stream."""
import os, re, subprocess, sys, zlib as pz
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import __graft_entry__ as ge
    z = ge.load(); z.init(0)
    comp = np.fromfile(sys.argv[2], dtype=np.uint8)
    out = z.inflate(comp)
    print("tier", z.last_inflate_tier(), len(out))
    sys.exit(0)

import __graft_entry__ as ge
import _oracle
z = ge.load()
for seed in (100, 101, 102, 103, 104, 105, 106, 107):
    raw = z.gen("itext", seed, 8 << 20)
    comp = np.frombuffer(pz.compress(raw.tobytes(), 6), dtype=np.uint8)
    path = os.path.join(ROOT, "gpurun_out", "t2c.bin")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    comp.tofile(path)
    env = dict(os.environ, ZES_T2_DBG="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=env, capture_output=True, text=True)
    m = re.search(r"zes T2 candidates buf 0 \((\d+)\):(.*)", r.stderr)
    cands = set(int(x) for x in m.group(2).split()) if m else set()
    starts, ends = _oracle.inflate_blocks(comp)
    missing = [s for s in starts[1:] if s not in cands]  # (the first block is work item 0 whether listed or not)
    print("seed %d: %d blocks, %d candidates, %d block starts not on the list: %s  %s" % (seed, len(starts), len(cands), len(missing), missing[:6], r.stdout.strip()), flush=True)
    for s in missing[:2]:
        k = starts.index(s)
        b = int.from_bytes(comp[s >> 3: (s >> 3) + 8].tobytes(), "little") >> (s & 7)
        print("   block %d at bit %d: bfinal %d btype %d hlit %d hdist %d hclen %d" % (k, s, b & 1, (b >> 1) & 3, ((b >> 3) & 31) + 257, ((b >> 8) & 31) + 1, ((b >> 13) & 15) + 4))
