#!/bin/bash
# the fp16x2 goldens (small cases and the full-size f_cfg2 / f_ek100, all against the reference's own outputs) under candidate one-pass site sets
for sites in "" "conv1d" "conv1d,linear.fc2" "conv1d,linear.fc1,linear.fc2"; do
  echo "=== AFFT_ONE_PASS_SITES='$sites'"
  AFFT_ONE_PASS_SITES="$sites" timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "fp16x2 and (forward_matches_reference_golden or (forward_full_size and default))" 2>&1 | grep -E "fp16x2|passed|failed|Error" | cut -c1-200
done
