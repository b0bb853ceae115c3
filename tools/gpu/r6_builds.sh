#!/bin/bash
# round 6: the build-differential fuzz, the fresh-model slice and the gate tests on the GPU box
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${1:-r6_builds}; mkdir -p $OUT
for seed in ${3:-601 602 603}; do
  timeout 1700 python3 tests/fuzz_builds.py ${2:-150} $seed > $OUT/fuzz_builds_$seed.txt 2>&1; echo "exit $?" >> $OUT/fuzz_builds_$seed.txt
  grep "^DIFF" $OUT/fuzz_builds_$seed.txt | head -20; tail -4 $OUT/fuzz_builds_$seed.txt
done
timeout 1700 python3 -m pytest tests/test_gpu_builds.py -x -q > $OUT/pytest_builds.txt 2>&1; tail -5 $OUT/pytest_builds.txt
