"""A fixed slice of the randomised parity fuzz (tests/fuzz_parity.py) in the suite: random (n, p, chains, kernel family, engine) cases in
both model dtypes and under both precision policies against the float64 oracle -- decisions, states, closures, chunk / shard / planned-
shard bit-exactness.  (Its first float64 run found a kernel that computed wrong states: profiles/r4_fuzz.txt.)  Longer campaigns:
`python tests/fuzz_parity.py 1000 <seed> <full|auto> <float32|float64>`."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cases,seed,precision,dtype", [(70, 101, "full", "float64"), (70, 102, "auto", "float64"), (70, 103, "full", "float32"),
                                                        (70, 104, "auto", "float32")])
def test_fuzz_slice(cases, seed, precision, dtype):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "fuzz_parity.py"), str(cases), str(seed), precision, dtype],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    tail = "\n".join(r.stdout.strip().splitlines()[-15:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    assert "0 failed" in r.stdout.strip().splitlines()[-1], tail
