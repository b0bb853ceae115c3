"""What the planner (logreg_amd/csrc/lr_plan.h) must choose at 256 CUs -- shared by the GPU test (through the C ABI:
tests/test_gpu_parity.py::test_planner_engine_choice_by_size) and the CPU test (tests/test_planner_cpu.py: the same planner code
as a host program, tests/host/plan_harness.hip).  One line per case:
    (n, p, chains, kernel family, precision policy, expectation)
expectation: a full plan {"mode", "group", "rows_per_lane"}, or {"mode": m}, or {"mode": m, "group": g}, or {"not_mode": m};
an optional "dtype": "float64" selects the float64 model."""


def REG(g, r):
    return {"mode": "reg", "group": g, "rows_per_lane": r}


def MFMA(s, r):
    return {"mode": "mfma", "group": s, "rows_per_lane": r}


CASES = [
    # ---- kernel-family-blind planning (MALA stands for "not HMC with reduced-precision interior steps")
    (200, 8, 4096, "mala", "auto", REG(16, 13)),
    # lanes per chain by the launch-time model: between the exactly-filled chain counts a wave alone on its SIMD beats two
    # narrower ones sharing it
    (200, 8, 1024, "mala", "auto", REG(64, 4)), (200, 8, 2048, "mala", "auto", REG(32, 7)), (200, 8, 2560, "mala", "auto", REG(16, 13)),
    (200, 8, 5120, "mala", "auto", REG(16, 13)), (200, 8, 64, "mala", "auto", {"mode": "reg", "group": 64}),
    # ... and beyond an exactly-filled count the run is planned in two parts: the filled head on 16 lanes per chain, the remainder
    # on the group width the model prefers for it (one short wave per SIMD); no second part where the model sees no gain
    (200, 8, 5120, "mala", "auto", {**REG(16, 13), "tail": {"from": 4096, "mode": "reg", "group": 64, "rows_per_lane": 4}}),
    (200, 8, 6144, "hmc", "full", {**REG(16, 13), "tail": {"from": 4096, "mode": "reg", "group": 32, "rows_per_lane": 7}}),
    (200, 8, 9216, "mala", "auto", {**REG(16, 13), "tail": {"from": 8192, "mode": "reg", "group": 64, "rows_per_lane": 4}}),
    (200, 8, 10240, "hmc", "full", {**REG(16, 13), "tail": {"from": 8192, "mode": "reg", "group": 32, "rows_per_lane": 7}}),
    (200, 8, 12288, "mala", "auto", {"no_tail": True}),
    (200, 8, 4096, "mala", "auto", {"no_tail": True}), (200, 8, 8192, "mala", "auto", {"no_tail": True}),
    (200, 8, 7168, "mala", "auto", {"no_tail": True}), (200, 8, 2560, "mala", "auto", {"no_tail": True}),
    (200, 8, 1 << 18, "mala", "auto", {"mode": "global", "group": 1}),
    (300, 8, 4096, "mala", "auto", REG(32, 16)), (600, 8, 4096, "mala", "auto", REG(64, 12)), (1000, 8, 4096, "mala", "auto", REG(64, 16)),
    (1500, 8, 4096, "mala", "auto", {"mode": "lds"}),
    (4000, 8, 1024, "mala", "auto", {"mode": "lds"}),        # 128 KB of rows, few chains
    (4000, 8, 2048, "mala", "auto", {"mode": "lds"}),        # (round 3: 64 lanes per chain still 1.6x the stepwise engine here)
    (4000, 8, 4096, "mala", "auto", {"mode": "stepwise"}),   # same rows, enough chains to fill the chip per slice
    (6000, 8, 64, "mala", "auto", {"mode": "stepwise"}),     # beyond LDS
    (300, 100, 64, "mala", "auto", {"mode": "stepwise"}),
    (200, 12, 2048, "mala", "auto", REG(32, 7)),
    # ---- HMC whose interior gradients may use the bf16 matrix pipe moves to the fused matrix-core kernels once there are enough
    # ---- chains; "full" precision never does
    (200, 8, 2048, "hmc", "auto", {"mode": "reg"}), (200, 8, 4096, "hmc", "auto", MFMA(4, 4)), (200, 8, 4096, "hmc", "full", REG(16, 13)),
    (200, 8, 8192, "hmc", "auto", MFMA(4, 4)), (200, 8, 16384, "hmc", "auto", MFMA(1, 13)), (200, 8, 16384, "hmc", "full", REG(16, 13)),
    # a remainder of at most a quarter of the exactly-filling count runs on a register kernel beside the matrix-core head (round 4)
    (200, 8, 5120, "hmc", "auto", {**MFMA(4, 4), "tail": {"from": 4096, "mode": "reg", "group": 64, "rows_per_lane": 4}}),
    (200, 8, 9216, "hmc", "auto", {**MFMA(4, 4), "tail": {"from": 8192, "mode": "reg", "group": 64, "rows_per_lane": 4}}),
    (200, 8, 18432, "hmc", "auto", {**MFMA(1, 13), "tail": {"from": 16384, "mode": "reg", "group": 32, "rows_per_lane": 7}}),
    (200, 8, 20480, "hmc", "auto", {**MFMA(1, 13), "tail": {"from": 16384, "mode": "reg", "group": 16, "rows_per_lane": 13}}),  # (runs after the head)
    (200, 8, 6144, "hmc", "auto", {**MFMA(4, 4), "no_tail": True}), (200, 8, 5120, "hmc", "bf16", {**MFMA(4, 4), "no_tail": True}),
    (200, 8, 16384, "mala", "auto", REG(16, 13)),
    # mid-size data: rows split over the 4 waves of a workgroup, 8 or 16 tiles per wave, from one workgroup per CU
    (700, 8, 4096, "hmc", "auto", MFMA(4, 16)), (700, 8, 2048, "hmc", "auto", {"mode": "reg"}),
    # wider models (9 <= p <= 32): the same kernel family from 4 chains per CU; no variant beyond 8 tiles per wave at p > 16
    (200, 12, 1024, "hmc", "auto", MFMA(4, 4)), (200, 12, 16384, "hmc", "auto", MFMA(1, 13)), (200, 12, 512, "hmc", "auto", {"mode": "reg"}),
    (200, 12, 4096, "hmc", "full", MFMA(4, 4)), (200, 12, 2048, "hmc", "full", {"mode": "reg"}),   # (round 4: the fp32 matrix-core kernel from 16 chains per CU)
    (500, 32, 4096, "hmc", "auto", MFMA(4, 8)), (900, 16, 4096, "hmc", "auto", MFMA(4, 16)),
    (900, 32, 4096, "hmc", "auto", MFMA(4, -1)),   # p > 16 beyond 8 tiles per wave: operands in device memory
    (900, 32, 1024, "hmc", "auto", MFMA(4, -1)),   # (p > 16, n <= 2048: from 4 chains per CU)
    (900, 32, 512, "hmc", "auto", {"not_mode": "mfma"}), (3000, 32, 1024, "hmc", "auto", {"not_mode": "mfma"}),
    # beyond the register variants: the same kernel with its bf16 operands in LDS, from one workgroup per CU
    (2000, 8, 4096, "hmc", "auto", MFMA(8, 0)), (2000, 8, 2048, "hmc", "auto", MFMA(8, 0)), (2000, 8, 1024, "hmc", "auto", {"not_mode": "mfma"}),
    (1150, 16, 4096, "hmc", "auto", {"mode": "mfma", "rows_per_lane": 0}),
    # beyond LDS: operand images in device memory
    (2600, 8, 4096, "hmc", "auto", MFMA(8, -1)), (2600, 8, 16384, "hmc", "auto", MFMA(4, -1)),
    (2600, 8, 2048, "hmc", "auto", MFMA(8, -1)),   # (round 3: from 8 chains per CU up to n = 6000)
    (2600, 8, 1024, "hmc", "auto", {"mode": "lds"}), (7000, 8, 2048, "hmc", "auto", {"mode": "stepwise"}),
    (3000, 16, 1024, "hmc", "auto", MFMA(4, -1)), (5000, 16, 1024, "hmc", "auto", {"mode": "stepwise"}),
    (9000, 8, 4096, "hmc", "auto", {"mode": "stepwise"}), (20000, 8, 4096, "hmc", "auto", {"mode": "stepwise"}),
    # float64 (p <= 8: rows in registers as well; wider: LDS / global only)
    (200, 8, 2048, "hmc", "full", {"dtype": "float64", "mode": "reg", "group": 32, "rows_per_lane": 7}),
    # ... from one wave per SIMD in LDS with 16 lanes per chain, from two with 8 (8 x 25 rows = 200 exactly; round 4)
    (200, 8, 4096, "hmc", "full", {"dtype": "float64", "mode": "lds", "group": 16}),
    (200, 8, 16384, "hmc", "full", {"dtype": "float64", "mode": "lds", "group": 8}), (200, 8, 65536, "hmc", "full", {"dtype": "float64", "mode": "lds", "group": 1}),
    (200, 8, 262144, "hmc", "full", {"dtype": "float64", "mode": "lds", "group": 1}), (200, 8, 65536, "mala", "full", {"dtype": "float64", "mode": "lds", "group": 8}),
    (200, 8, 8192, "mala", "auto", {"dtype": "float64", "mode": "lds", "group": 8}),
    (200, 12, 4096, "hmc", "full", {"dtype": "float64", "mode": "lds", "group": 8}), (200, 12, 2048, "mala", "auto", {"dtype": "float64", "mode": "lds", "group": 64}),
    (200, 3, 4096, "mala", "auto", {"dtype": "float64", "mode": "lds", "group": 8}),
    # float64 HMC under the default precision policy, 5 <= p <= 8, n <= 256: float32 interior gradients (k_chain_mixed)
    (200, 8, 4096, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 16, "rows_per_lane": 13}),
    (200, 8, 64, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 64, "rows_per_lane": 4}),  # few chains: wide lane groups
    (200, 8, 2048, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 32, "rows_per_lane": 7}),
    (256, 8, 512, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 64, "rows_per_lane": 4}),
    (250, 6, 8192, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 16, "rows_per_lane": 16, "no_tail": True}),
    (200, 8, 10240, "hmc", "auto", {"dtype": "float64", "mode": "mfma", "group": 1, "rows_per_lane": 13, "no_tail": True}),  # k_chain_mfma_f64
    (200, 8, 18432, "hmc", "auto", {"dtype": "float64", "mode": "mfma", "group": 1, "rows_per_lane": 13,
                                     "tail": {"from": 16384, "mode": "mixed", "group": 32, "rows_per_lane": 7}}),
    (200, 8, 24576, "hmc", "auto", {"dtype": "float64", "mode": "mfma", "group": 1, "no_tail": True}),
    (200, 8, 8448, "hmc", "auto", {"dtype": "float64", "mode": "mfma", "group": 1}), (200, 8, 8447, "hmc", "auto", {"dtype": "float64", "mode": "mixed"}),
    # ... planned in two parts between exactly-filled chain counts, as the register family is (the parts run in turn)
    (200, 8, 5120, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 16, "rows_per_lane": 13,
                                    "tail": {"from": 4096, "mode": "mixed", "group": 64, "rows_per_lane": 4}}),
    (200, 8, 6144, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 16, "rows_per_lane": 13,
                                    "tail": {"from": 4096, "mode": "mixed", "group": 32, "rows_per_lane": 7}}),
    (200, 8, 7168, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "no_tail": True}),
    (200, 8, 4096, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "no_tail": True}),
    (250, 8, 1 << 16, "hmc", "auto", {"dtype": "float64", "mode": "mixed"}),  # 250 rows: beyond the 13 register tiles
    (300, 8, 4096, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 32, "rows_per_lane": 16}),  # (replicated-state form)
    (1000, 8, 4096, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 64, "rows_per_lane": 16}),
    (1100, 8, 4096, "hmc", "auto", {"dtype": "float64", "not_mode": "mixed"}),
    (200, 4, 4096, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 16, "rows_per_lane": 16}),
    (200, 12, 4096, "hmc", "auto", {"dtype": "float64", "mode": "mixed", "group": 32, "rows_per_lane": 7}),
    (200, 24, 4096, "hmc", "auto", {"dtype": "float64", "not_mode": "mixed"}),
    (200, 8, 4096, "mala", "auto", {"dtype": "float64", "not_mode": "mixed"}),
    # float64 wide models: the stepwise engine on the f64 matrix pipe (lr_wide_f64.h), 64 chains per workgroup
    (4096, 128, 1024, "hmc", "auto", {"dtype": "float64", "mode": "stepwise", "group": 16}),
    (300, 100, 64, "mala", "auto", {"dtype": "float64", "mode": "stepwise"}),
    # rows beyond the scalar cache, very many chains: the register tier keeps its rank whatever the modelled cost (the
    # unbounded model term let lds / global variants overtake from ~98 304 chains, with no measurement behind the flip)
    (1000, 8, 1 << 17, "mala", "auto", {"mode": "reg"}), (1000, 8, 1 << 19, "hmc", "full", {"mode": "reg"}),
    # one chain per wave in registers and many chains (round 4, profiles/r4_planner_bench_many_chains.txt): 9 <= p <= 16 moves to the
    # fp32 matrix-core kernel from 16 chains per CU in all-fp32 arithmetic too; rows within 28 KB to LDS with 8 lanes per chain from 64
    (500, 16, 4096, "mala", "auto", MFMA(4, 8)), (500, 16, 4096, "hmc", "full", MFMA(4, 8)), (500, 16, 2048, "mala", "auto", REG(64, 8)),
    (300, 12, 8192, "mala", "auto", MFMA(4, 8)), (300, 12, 16384, "mala", "auto", {"mode": "lds", "group": 8}),
    (500, 16, 1 << 16, "mala", "auto", MFMA(4, 8)),   # 32 KB of rows: not LDS
    (600, 8, 16384, "mala", "auto", {"mode": "lds", "group": 8}), (600, 8, 8192, "mala", "auto", REG(64, 12)),
    (200, 12, 4096, "mala", "auto", MFMA(4, 4)), (200, 24, 4096, "hmc", "full", MFMA(4, 4)), (400, 30, 4096, "mala", "auto", MFMA(4, 8)),
    (200, 24, 1 << 15, "mala", "auto", MFMA(4, 4)), (900, 16, 4096, "mala", "auto", MFMA(4, 16)), (900, 30, 4096, "mala", "auto", {"not_mode": "mfma"}),
    (800, 8, 1 << 17, "hmc", "full", {"mode": "lds", "group": 8}), (400, 8, 16384, "mala", "auto", REG(32, 16)),
]


def matches(plan: dict, expect: dict) -> bool:
    if "not_mode" in expect:
        return plan["mode"] != expect["not_mode"]
    if expect.get("no_tail") and "tail" in plan:
        return False
    return all(plan.get(k) == v for k, v in expect.items() if k not in ("dtype", "no_tail"))
