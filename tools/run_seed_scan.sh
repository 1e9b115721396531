#!/bin/bash
# kink-free seeds for tests/test_gpu_model.py::test_gradient_parity_tight_on_kink_free_seeds: every case on every arithmetic route (N C H W depth automask)
for c in "1 1 64 96 18 0" "2 1 64 96 18 0" "2 3 64 96 18 1" "3 1 96 160 18 0" "5 3 64 64 18 0" "1 1 64 256 18 0" "2 3 64 256 18 0"; do
  python3 tools/seed_scan.py $c 100 ${1:-16} || exit 1
done
