"""Runs the malformed-stream cases one by one, printing each name first (finds a case that hangs; not a pytest)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
z = ge.load(); z.init(0)
cases = json.load(open(os.path.join(ROOT, "tests", "golden", "inflate_cases.json")))
for e in cases:
    print(e["name"], len(e["input"]) // 2, flush=True)
    try:
        got = ("out", z.inflate(bytes.fromhex(e["input"])).tobytes().hex())
    except z.ZlibEsError as ex:
        got = ("err", str(ex))
    exp = ("err", e["error"]) if "error" in e else ("out", e["output"])
    print("   ", "ok" if got == exp else "MISMATCH", "tier", z.last_inflate_tier(), flush=True)
