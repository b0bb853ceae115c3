#!/usr/bin/env python3
"""Fresh-model stress campaign (the scenarios and the probe live in tests/stress_fresh_models.py; the suite runs a slice of it):
    python3 tools/fresh_model_stress.py [reps]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import logreg_amd as la
from stress_fresh_models import SCEN, run_scenario

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for scen in SCEN:
    bad, runs, plan = run_scenario(la, scen, reps)
    name, opt, dtype, n, p, C, L = scen
    print(f"{name}: {bad} of {runs} runs differ from the first  [{dtype}, n={n}, p={p}, {C} chains, L={L}, plan {plan}, opts '{opt}']", flush=True)
