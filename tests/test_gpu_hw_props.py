"""Hardware behaviour the kernels rest on, checked on the box the tests run on.

k_lz_sort ranks the elements of a digit with ONE returning LDS add per element (zes_deflate.hip, the scatter passes): that
is the stable rank only if lanes of one wavefront that meet in a counter word are served in ascending lane order.  The
architecture manual does not promise it; gfx950 does it (tools/micro/lds_atomic_order.hip: 0 of 3.3e10 values out of
order).  A device that served them otherwise would still produce valid DEFLATE streams, but not the reference's bytes —
the parity tests would show that too; this test names the cause."""
import pytest

pytestmark = pytest.mark.gpu


def test_returning_lds_adds_of_one_wavefront_are_served_in_lane_order(gpu, z):
    total = 0
    for seed in (1, 7919, 0xC0FFEE):
        bad, n = z.selftest_lds_order(iters=300, seed=seed)
        assert bad == 0, f"{bad} of {n} values out of lane order (seed {seed})"
        total += n
    assert total > 9 * 10 ** 8  # 256 workgroups x 1024 lanes x 300 rounds x 4 adds per seed, minus the holes
