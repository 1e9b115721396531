#!/bin/bash
# kink-free seeds for tests/test_gpu_model.py::test_gradient_parity_tight_on_kink_free_seeds: every case in deterministic AND default mode (N C H W depth automask)
for c in "1 1 64 96 18 0" "2 1 64 96 18 0" "2 3 64 96 18 1" "3 1 96 160 18 0" "5 3 64 64 18 0" "1 1 64 256 18 0" "2 3 64 256 18 0"; do
  MD2_DETERMINISTIC=1 python3 tools/seed_scan.py $c 100 12 | sed 's/^/det /'
  python3 tools/seed_scan.py $c 100 12 | sed 's/^/def /'
done
