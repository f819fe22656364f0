"""Stage-level LZ77 (zes_stage_lz77_dev) against the oracle on a few inputs (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import _oracle
z = ge.load(); z.init(0)
for name, a in (("zeros 90000", np.zeros(90000, dtype=np.uint8)), ("zeros 131072", np.zeros(131072, dtype=np.uint8)),
                ("period 7", np.resize(np.arange(7, dtype=np.uint8), 100000)), ("random", z.gen("xorshift", 3, 100000)), ("text", z.gen("itext", 3, 100000))):
    for order in (0, 1):
        got = z.stage_lz77_tensor(torch.from_numpy(a).cuda(), 0, len(a))
        want = _oracle.lz77_block(a, 0, len(a))
        ok = len(got) == len(want) and bool((got == want).all())
        first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
        print(name, "run", order, "tokens", len(got), "want", len(want), "ok", ok, "first diff", first, (hex(int(got[first])), hex(int(want[first]))) if first is not None else "", flush=True)
