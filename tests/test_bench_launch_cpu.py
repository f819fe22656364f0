"""`python bench.py --gpus N` (N > 1) without a launcher: the process must start the ranks as a CHILD process
(torch.distributed.run, the driver's own command shape) before anything touches the GPU — here: before torch is
even imported — relay rank 0's JSON line and the exit code, and fall back to --no-gather once if the ranks fail."""
import json
import os
import subprocess
import sys

from conftest import ROOT

DRIVER = r'''
import json, subprocess, sys
sys.path.insert(0, %(root)r)
import bench
calls = []
plan = json.loads(%(plan)r)
def fake_run(cmd, **kw):
    assert "stdout" in kw and "stderr" in kw and kw.get("env", {}).get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    calls.append(cmd)
    rc, out = plan[len(calls) - 1]
    return subprocess.CompletedProcess(cmd, rc, stdout=out)
subprocess.run = fake_run
sys.argv = ["bench.py"] + %(argv)r
try:
    bench.main()
    rc = 0
except SystemExit as e:
    rc = e.code
assert "torch" not in sys.modules, "the parent imported torch before starting the ranks"
print("CALLS " + json.dumps({"rc": rc, "calls": calls}))
'''


def _run(argv, plan):
    code = DRIVER % {"root": ROOT, "plan": json.dumps(plan), "argv": argv}
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("CALLS ")][0][6:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return rec, lines


def test_gpus_n_starts_the_ranks_as_a_child_and_relays_the_line():
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 2, "config": {"gather": "x"}})
    rec, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], [[0, "noise\n" + line + "\n"]])
    assert rec["rc"] == 0 and len(rec["calls"]) == 1 and lines == [line]
    cmd = rec["calls"][0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]


def test_failed_ranks_are_started_once_more_without_the_gather():
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 4, "config": {"gather": None}})
    rec, lines = _run(["--gpus", "4"], [[1, "Traceback ...\n"], [0, line + "\n"]])
    # the second launch's measurement is relayed, but the first launch's failure is neither hidden in the line nor in the exit code
    assert rec["rc"] == 3 and len(rec["calls"]) == 2
    assert "--no-gather" not in rec["calls"][0] and rec["calls"][1][-1] == "--no-gather"
    d = json.loads(lines[0])
    assert len(lines) == 1 and "FAILED" in d["config"]["gather"] and d["first_launch_failed"] == 1 and "first_launch_stderr_tail" in d


def test_failure_of_both_launches_is_reported():
    rec, lines = _run(["--gpus", "2"], [[3, ""], [5, ""]])
    assert rec["rc"] == 5 and lines == [] and len(rec["calls"]) == 2


def test_inproc_mode_plans_the_configs3_buffers_and_their_owners(z):
    """`bench.py --gpus N --inproc` (one process, N device contexts, host batch API): its buffers are BASELINE configs[3]'s —
    buffer i = generator i % 3, seed 12345 + i, 1 MiB, the very entries of the reference-run golden file — 128 per GPU, and
    the library's partition rule (zes_partition, no GPU needed) deals them out evenly."""
    sys.path.insert(0, ROOT)
    import bench

    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "batch1m.json")))
    for n in (1, 2, 8):
        plan = bench.inproc_plan(n)
        assert len(plan) == 128 * n
        for i, (kind, seed, size) in enumerate(plan):
            assert (kind, seed, size) == (gold[i]["kind"], gold[i]["seed"], gold[i]["n"]) and gold[i]["i"] == i
        own = z.partition([p[2] for p in plan], n)
        assert [own.count(d) for d in range(n)] == [128] * n
