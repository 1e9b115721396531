#!/bin/bash
# bf16-tensor configurations, tree library against another build (same box): tools/run_bs_ab.sh <other.so>
AB_ARMS="1:-1" bash tools/ab_lib.sh "$1" 8,18,3,320,1024,bf16_storage 16,50,3,192,640,bf16_storage 32,18,1,128,416,bf16_storage 4,18,1,128,416,bf16_storage
