"""A fixed-seed slice of tools/gpu_fuzz.py in the suite: a few hundred randomised cases — sizes, generators, runs and
short periods, settings of another DEFLATE encoder (level, memLevel, window, strategy, flushes), bit flips, overwritten
spans and truncations, views with live bytes behind them, batches with damaged members — each compared with the
oracle (bytes, or which of the reference's errors: src/inflate.ts:22-37, src/utils/BitReadStream.ts:14-42).
The long soaks of that tool found every real bug of rounds 1 and 2; this keeps their surface under the driver's run."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,damage,cases", [(77, "1", 160), (910, "2", 120), (3001, "2", 120)])
def test_fuzz_slice_matches_the_oracle(gpu, seed, damage, cases):
    env = dict(os.environ, FUZZ_MAX_CASES=str(cases), FUZZ_DAMAGE=damage)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz.py"), "240", str(seed)], capture_output=True, text=True,
                         timeout=400, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "fuzz ok: %d cases" % cases in out.stdout, out.stdout[-500:]
