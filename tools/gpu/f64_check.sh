#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-f64_check}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "float64_at_padded_width_32" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for spec in "400 501 full float64" "300 503 auto float64"; do
  FUZZ_P="17,20,24,31,32" timeout 900 python3 tests/fuzz_parity.py $spec > $OUT/fuzz_$(echo $spec | tr ' ' '_').txt 2>&1
  echo "fuzz $spec: $(tail -1 $OUT/fuzz_$(echo $spec | tr ' ' '_').txt)"; grep "^FAIL" $OUT/fuzz_$(echo $spec | tr ' ' '_').txt | head -5
done
bash tools/gpu/f64_repro.sh $1 > /dev/null 2>&1; cat $OUT/repro.txt | cut -c1-330
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/sigmoid_probe.hip -o /tmp/sigmoid_probe && /tmp/sigmoid_probe > $OUT/sigmoid_probe.txt 2>&1; cat $OUT/sigmoid_probe.txt
