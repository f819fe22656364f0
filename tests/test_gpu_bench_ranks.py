"""`python bench.py --gpus 2` end to end on a one-GPU box: the parent starts the ranks as a child process
(torch.distributed.run), every rank deflates and inflates its own buffer (seed 12345 + rank) and checks it against the
reference-run golden of ITS seed, the compressed shards are gathered on rank 0 (first step: sizes first; then from the
lengths of the step before), the times are reduced, rank 0 prints the line.  The ranks share the GPU and talk over gloo
(ZES_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device) — a rehearsal of every line of the N > 1 path but the
transport, never a measurement."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_rehearsal_over_gloo(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ZES_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["verified_bit_exact"] is True and d["golden_sha256_checked"] is True  # rank 1's seed has its own golden
    assert d["text64"]["verified_bit_exact"] is True and d["text64"]["golden_sha256_checked"] is True
    assert "rehearsal" in d["config"] and "FAILED" not in d["config"]["gather"]
    got = int(d["config"]["gather"].split("region: ")[1].split(" ")[0])
    assert got == 2 * d["config"]["compressed_bytes"] or abs(got - 2 * d["config"]["compressed_bytes"]) < 4096  # both ranks' shards reached rank 0
