"""The planner of the C ABI (logreg_amd/csrc/lr_plan.h) in the GPU-less build container: tests/host/plan_harness.hip compiles the
very planning code of the library as a host program (no HIP call; linked against the library's own instantiation objects, so the
variant tables are the real ones) and answers plan requests on stdin.  The expectations are the ones the GPU test checks through
the C ABI (tests/planner_cases.py): host-side planning defects -- a threshold, an LDS-size formula, a variant that does not fit --
show here, not in metered GPU minutes."""
import os
import subprocess

import pytest

from conftest import REPO
from planner_cases import CASES, matches

MODES = {0: "reg", 1: "lds", 2: "global", 3: "mfma", 4: "stepwise", 5: "mixed"}
KIND = {"rwmh": 0, "mala": 1, "hmc": 2, "ul": 3}
PREC = {"auto": 0, "full": 1, "bf16": 2}


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    from logreg_amd import build as b
    b.build(verbose=False)  # the instantiation objects the harness links against
    out = tmp_path_factory.mktemp("plan_harness")
    obj, exe = str(out / "plan_harness.o"), str(out / "plan_harness")
    hipcc = b._hipcc()
    inc = ["-I", os.path.join(REPO, "logreg_amd", "csrc"), "-I", os.path.join(REPO, "include")]
    # host code under AddressSanitizer + UBSan (host side only: -Xarch_host; a finding aborts the harness and fails the test)
    san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-sanitize-recover=undefined"]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", *san, *inc, "-c", os.path.join(REPO, "tests", "host", "plan_harness.hip"),
                    "-o", obj], check=True, capture_output=True)
    objs = sorted(os.path.join(b.OBJDIR, f) for f in os.listdir(b.OBJDIR) if f.startswith("lr_inst_") and f.endswith(".o"))
    assert len(objs) == 12
    subprocess.run([hipcc, "--offload-arch=gfx950", *san[:2], obj, *objs, "-o", exe], check=True, capture_output=True)

    def ask(requests, cus=256):
        """requests: (dtype, p, n, chains, kind, precision, group, mode) -> plan dicts or ("ERR", text)"""
        text = "".join(f"{d} {p} {n} {c} {KIND[k]} {PREC[pr]} {g} {m} {cus}\n" for d, p, n, c, k, pr, g, m in requests)
        r = subprocess.run([exe], input=text, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0, r.stderr[-3000:]
        res = []
        for line in r.stdout.strip().split("\n"):
            f = line.split()
            if f[0] == "ERR":
                res.append(("ERR", line))
                continue
            plan = {"mode": MODES[int(f[0])], "group": int(f[1]), "rows_per_lane": int(f[2]), "lds_bytes": int(f[3]), "interior": (int(f[4]), int(f[5]), int(f[6]))}
            if int(f[7]) > 0:
                plan["tail"] = {"from": int(f[7]), "mode": MODES[int(f[10])], "group": int(f[8]), "rows_per_lane": int(f[9])}
            res.append(plan)
        return res
    ask.exe = exe
    return ask


def test_planner_choices_on_the_cpu(harness):
    reqs = [(1 if e.get("dtype") == "float64" else 0, p, n, C, kind, prec, 0, -1) for n, p, C, kind, prec, e in CASES]
    plans = harness(reqs)
    bad = [(c, pl) for c, pl in zip(CASES, plans) if isinstance(pl, tuple) or not matches(pl, c[5])]
    assert not bad, bad


def test_planner_scales_with_the_cu_count_and_sizes_lds_within_the_chip(harness):
    # thresholds are in chains per CU: a chip of 128 CUs moves HMC to the matrix-core kernel at half the chain counts
    a, b, c = harness([(0, 8, 200, 2048, "hmc", "auto", 0, -1), (0, 8, 200, 1024, "hmc", "auto", 0, -1), (0, 8, 200, 5120, "hmc", "auto", 0, -1)], cus=128)
    assert a["mode"] == "mfma" and a["group"] == 4 and b["mode"] == "reg"
    assert {k: c[k] for k in ("mode", "group", "rows_per_lane", "lds_bytes")} == {"mode": "mfma", "group": 1, "rows_per_lane": 13, "lds_bytes": 0}
    # every LDS-resident plan fits the 160 KB of a CU, for every row count up to what the variant accepts (odd tile counts per
    # wave included: round 2 found its LDS-size defect on the GPU)
    reqs = [(0, p, n, C, "hmc", "auto", 0, -1) for p in (8, 12) for n in range(1040, 2600, 37) for C in (2048, 4096, 32768)]
    for rq, pl in zip(reqs, harness(reqs)):
        assert not isinstance(pl, tuple), (rq, pl)
        assert pl["lds_bytes"] <= 160 * 1024 - 4096, (rq, pl)
        if pl["mode"] == "mfma" and pl["rows_per_lane"] == 0:
            assert pl["lds_bytes"] > 0 and pl["group"] in (1, 4, 8)


def test_float64_default_policy_plans_size_their_lds_on_the_cpu(harness):
    """float64 models, HMC under the default policy, over a sweep of rows, widths and chain counts: the float32-interior kernels ask for
    their float64 rows + the per-lane stash, the matrix-core kernel for the rows (and for the stash as well when a remainder runs on
    the float32-interior kernels behind it), everything within a CU's 160 KB; 17 <= p <= 32 runs all-float64 on the distributed-state kernel (16 or 64 lanes per chain) while the
    rows fit the LDS."""
    reqs = [(1, p, n, C, "hmc", "auto", 0, -1) for p in (3, 8, 12, 16, 20) for n in (1, 100, 200, 208, 209, 256, 257, 512, 1000, 1024, 1025, 2000)
            for C in (1, 1000, 4096, 5120, 8448, 16384, 18432, 65536)]
    for rq, pl in zip(reqs, harness(reqs)):
        assert not isinstance(pl, tuple), (rq, pl)
        _, p, n, C = rq[:4]
        P = 4 if p <= 4 else 8 if p <= 8 else 16 if p <= 16 else 32
        rows, padded = n * P * 8, n * (P + 2) * 8  # (float64 rows in LDS: two doubles of padding per row against bank conflicts)
        if P == 32:
            assert pl["mode"] == "stepwise" or (pl["mode"] in ("lds", "global") and pl["group"] in (16, 64) and "tail" not in pl), (rq, pl)
            if pl["mode"] == "lds":
                assert pl["lds_bytes"] == padded <= 160 * 1024, (rq, pl)
        elif pl["mode"] == "mixed":
            assert pl["group"] * pl["rows_per_lane"] >= n and pl["lds_bytes"] == padded + 16 * 8 * 256 <= 160 * 1024, (rq, pl)
        elif pl["mode"] == "mfma":
            assert P == 8 and n <= 208 and C >= 33 * 256 and pl["group"] == 1, (rq, pl)
            assert pl["lds_bytes"] == (padded + 16 * 8 * 256 if "tail" in pl else rows), (rq, pl)
            if "tail" in pl:
                assert pl["tail"]["mode"] == "mixed", (rq, pl)
        else:  # beyond the register shapes: the all-float64 kernels
            assert n > 256 or P == 32 or (P == 16 and n > 512) or (P == 4 and n > 1024) or (P == 8 and n > 1024), (rq, pl)
            assert pl["lds_bytes"] <= 160 * 1024, (rq, pl)


def test_group_is_validated_per_mode_on_the_cpu(harness):
    res = harness([(0, 8, 200, 100, "mala", "auto", 48, -1), (0, 8, 200, 100, "hmc", "auto", 2, 3), (0, 8, 20000, 1024, "hmc", "auto", 63, 4),
                   (0, 8, 20000, 1024, "hmc", "auto", 10 ** 6, 4), (0, 100, 300, 64, "hmc", "auto", 5, -1)])
    assert res[0][0] == "ERR" and res[1][0] == "ERR" and res[3][0] == "ERR"
    assert res[2]["mode"] == "stepwise" and res[2]["group"] == 63  # any slice count up to one per 32-row block
    assert res[4]["mode"] == "stepwise" and res[4]["group"] == 5


def test_interior_step_slicing_on_the_cpu(harness):
    """lr_plan.h plan_interior: the row slices of the reduced-precision interior leapfrog steps (BASELINE configs 4 and 5 and
    the shapes around the 16-wave kernel's conditions)."""
    shapes = [(8, 100000, 1024), (128, 4096, 1024), (128, 4096, 8192), (30, 5000, 1024), (12, 20000, 1024), (8, 100000, 4096), (8, 3000, 1024),
              (100, 300, 64), (8, 200, 4096)]
    got = [pl["interior"] for pl in harness([(0, p, n, C, "hmc", "auto", 0, 4) for p, n, C in shapes])]
    cfg4, cfg5, cfg5_all, p30, p12, cfg4_4096, small_tall, small_wide, pima = got
    assert cfg4 == (16, 6272, 16)          # config 4: 16 slices x 16 chain blocks = one 16-wave workgroup per CU
    assert cfg5 == (4, 1024, 8)            # config 5 at one GPU's 1024 chains: 64 chain tiles x 4 row slices of 8 waves
    assert cfg5_all[0] == 0                # 8192 chains: more chain tiles than CUs -> no row split (the trajectory kernel's range)
    assert p30[2] == 4 and p30[0] * p30[1] >= 5000          # 5 slices of the 16-wave form would leave the chip two-thirds empty
    assert p12[2] == 16 and p12[0] <= 16 and p12[0] * p12[1] >= 20000
    assert cfg4_4096[2] == 4               # 4096 chains: the saved launch no longer shows, 4-wave workgroups
    assert small_tall[0] >= 1 and small_tall[1] >= 256 and small_tall[1] % 32 == 0
    assert small_wide == (1, 512, 8) or (small_wide[0] * small_wide[1] >= 300 and small_wide[1] % 256 == 0)
    assert pima[0] == 0                    # no stepwise operand images for Pima-size data
    # every slicing covers all rows with whole 32-row tile pairs, for a sweep of row counts
    reqs = [(0, p, n, C, "hmc", "auto", 0, 4) for p in (8, 16, 128) for n in range(4100, 120000, 7919) for C in (64, 1024, 2048) if p <= 32 or n <= 20000]
    for rq, pl in zip(reqs, harness(reqs)):
        rs, ln, wv = pl["interior"]
        if rs:
            assert rs * ln >= rq[2] and (rs - 1) * ln < rq[2] and ln % 32 == 0 and wv in (4, 8, 16), (rq, pl)


def test_half_precision_image_rounding_and_range_rule(harness):
    """The f16 one-piece image of the trajectory kernels (lr_wide_bf16.h wide_f16_prepare_rne): its float32 -> binary16 conversion is
    numpy's round-to-nearest-even bit for bit (normals, subnormals, ties, the zero threshold), and the image is refused unless every
    |x| <= 2^15 and every non-zero column reaches 2^-10."""
    import numpy as np
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(4000) * np.exp2(rng.integers(-28, 15, 4000)), [0.0, -0.0, 2.0**-24, 2.0**-25, 2.0**-25 * 1.0000001, 3 * 2.0**-25,
                        2.0**-14, 2.0**-14 * (1 - 2.0**-12), 32768.0, -32768.0, 1 + 2.0**-11, 1 + 3 * 2.0**-11, 1 + 2.0**-11 + 2.0**-20, 65504.0 / 2]]).astype(np.float32)
    x = x[np.abs(x) <= 32768]
    r = subprocess.run([harness.exe, "f16"], input="".join("%08x\n" % u for u in x.view(np.uint32)), capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.array([int(t, 16) for t in r.stdout.split()], dtype=np.uint16)
    assert np.array_equal(got, x.astype(np.float16).view(np.uint16))

    def fit(rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        rr = subprocess.run([harness.exe, "f16fit"], input="%d\n" % len(rows) + "".join("%08x\n" % u for u in rows.view(np.uint32).ravel()), capture_output=True, text=True, env=env)
        assert rr.returncode == 0, rr.stderr[-2000:]
        return rr.stdout.split()
    rows = rng.standard_normal((70, 64)).astype(np.float32)
    rows[:, 60:] = 0  # padding columns are all zero: allowed
    ok = fit(rows)
    assert ok[0] == "1" and int(ok[1]) == int(rows.astype(np.float16).view(np.uint16).astype(np.uint64).sum())  # every element placed exactly once, the rest zero
    big = rows.copy(); big[3, 7] = 40000.0
    tiny = rows.copy(); tiny[:, 9] *= 1e-5
    assert fit(big) == ["0"] and fit(tiny) == ["0"]
