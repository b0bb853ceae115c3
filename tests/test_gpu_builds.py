"""Trust by construction (round 6): two compilations of the same sources must agree bit for bit, and a fresh model's first run must
be the run every later one repeats.

Both wrong-result kernels this project ever shipped were invisible to rerun-determinism and to the size of stress the suite ran:
round 4's float64 p = 32 MALA kernel was a register-allocator copy placed ahead of a join block's EXEC restore under
-amdgpu-sched-strategy=max-ilp (profiles/r6_f64_p32_bisect.txt; logreg_amd/isa_gate.py now refuses the pattern at build time), round
5's two-tile trajectory kernel a hand-counted wait the compiler undid.  The first kind changes with the compiler's flags, the second
with timing: hence a SECOND BUILD of the library (logreg_amd/lib_alt: default scheduler, SLP on, no loop alignment, AGPR-form MFMA --
no fast-math in either, so scheduling and allocation may not change a result) run through the same fuzz cases, and a fresh-model slice
of every kernel family with hand-managed synchronisation."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import logreg_amd
    return logreg_amd


@pytest.mark.parametrize("cases,seed", [(400, 601), (400, 602)])
def test_two_builds_of_the_same_sources_agree_bit_for_bit(cases, seed):
    """800 random cases (both dtypes, both precision policies, all four kernel families, every engine the planner can be forced onto,
    few and many chains, p up to 128) through liblogreg_hip.so and through lib_alt/liblogreg_hip.so: samples, final states, accept
    counts and closure values identical to the last bit."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "fuzz_builds.py"), str(cases), str(seed)],
                       capture_output=True, text=True, timeout=1500, cwd=REPO)
    lines = r.stdout.strip().splitlines()
    tail = "\n".join(lines[-15:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    assert " 0 differ" in lines[-1], tail
    assert int(lines[-1].split(":")[1].split("cases run")[0]) >= 0.8 * cases, tail  # (skips: engines the shape does not admit)


def test_fresh_models_repeat_their_first_run(la):
    """20 fresh models per scenario of tests/stress_fresh_models.py (row-split / chain-split / trajectory kernels of the wide engine,
    both tall interior kernels, the matrix-core chain kernel with operands in registers and in LDS, the register kernel; float32 and
    float64 states): the seeded run and its chunked repeat on every one of them bit-identical with the first model's run."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from stress_fresh_models import SCEN, run_scenario
    bad = []
    for scen in SCEN:
        b, runs, plan = run_scenario(la, scen, 20)
        if b:
            bad.append((scen[0], b, runs, plan))
    assert not bad, bad
