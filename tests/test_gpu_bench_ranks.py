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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["verified_bit_exact"] is True and d["golden_sha256_checked"] is True  # rank 1's seed has its own golden
    assert d["text64"]["verified_bit_exact"] is True and d["text64"]["golden_sha256_checked"] is True
    assert "rehearsal" in d["config"] and "FAILED" not in d["config"]["gather"]
    assert d["ranks"]["world_size"] == 2 and d["ranks"]["gathered_bytes_per_step"] > 0 and d["repeats"] == 3
    assert d["value_spread"]["min"] <= d["value"] <= d["value_spread"]["max"]
    got = int(d["config"]["gather"].split("region: ")[1].split(" ")[0])
    assert got == 2 * d["config"]["compressed_bytes"] or abs(got - 2 * d["config"]["compressed_bytes"]) < 4096  # both ranks' shards reached rank 0


def test_one_process_two_contexts_inproc_rehearsal(gpu):
    """`bench.py --gpus 2 --inproc`: ONE process, zes_init_devices(2) (both contexts on this box's one GPU: ZES_OVERSUBSCRIBE),
    256 x 1 MiB of configs[3] from pinned host arrays through the host batch forms, every buffer against the reference-run
    golden — the road to a scaling point that does not pass through RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ZES_OVERSUBSCRIBE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--inproc", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["buffers_per_step"] == 256 and d["config"]["buffers_per_context"] == [128, 128]
    assert d["verified_bit_exact"] is True and d["golden_sha256_checked"] is True and d["config"]["oversubscribed"] is True
    assert d["value"] > 0 and d["deflate_gibs"] > 0 and d["inflate_gibs"] > 0


def test_every_leg_of_the_default_line(gpu):
    """`python bench.py` (what the driver runs, shortened): the main leg and every further leg — text64, batch1m, lowent256,
    zlibtext64, host_api — are in the ONE line, each verified, each with its roofline."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["verified_bit_exact"] is True and d["golden_sha256_checked"] is True and d["roofline"]["frac"] > 0 and d["pool_bytes"] > 0
    for name in ("text64", "batch1m", "lowent256", "zlibtext64"):
        leg = d[name]
        assert leg["verified_bit_exact"] is True and leg["roofline"]["frac"] > 0 and leg["inflate_gibs_per_gpu"] > 0, name
        if name != "zlibtext64":
            assert leg["golden_sha256_checked"] is True and leg["deflate_gibs_per_gpu"] > 0 and leg["roofline_deflate"]["frac"] > 0, name
    h = d["host_api"]
    assert h["verified_bit_exact"] is True and set(h["rows"]) == {"random64_pageable", "random64_pinned", "text64_pageable", "text64_pinned"}
    for r in h["rows"].values():
        assert r["golden_sha256_checked"] is True and r["deflate_gibs"] > 0 and r["inflate_gibs"] > 0
    assert "failed" not in (h["node"] or {})
