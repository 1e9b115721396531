#!/bin/bash
# Round-3 profile set, run on the GPU box from the repo root (tools/make_profiles.sh); raw output under gpurun_out/prof_final/, summaries by tools/profile_summary.py.
# One rocprofv3 pass per purpose: kernel trace + stats (both stream modes, batch 32 and batch 4), then the two HBM counters in separate --pmc passes (never with other trace
# domains), then the vector-instruction counter of the fused loss kernel (its binding roofline).
set -e
R=$PWD; O=$R/gpurun_out/prof_final; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
MD2_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/one.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/two -o t -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $O/two.log 2>&1
MD2_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b4 -o t -- python3 $R/tools/prof_config.py 4 18 1 128 416 f32 10 > $O/b4.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/valu -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/valu.log 2>&1
# configuration #5 (1024x320, bf16 mixed precision, batch 8): HBM traffic of its kernel classes for the roofline block bench.py prints with that run
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch5 -o t -- python3 $R/tools/prof_config.py 8 18 3 320 1024 bf16 3 > $O/fetch5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write5 -o t -- python3 $R/tools/prof_config.py 8 18 3 320 1024 bf16 3 > $O/write5.log 2>&1
cd $R
tail -1 $O/one.log | cut -c1-300; tail -1 $O/two.log | cut -c1-300; grep ms/step $O/b4.log
python3 tools/profile_summary.py stats $O/one 10 "rocprofv3 --kernel-trace --stats summary, round 3: MD2_ONE_STREAM=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra (B=32; 10 steps in the trace: 2 warm-up + 5 timed + 3 profiled). Every kernel alone: the mode the bench roofline is timed in." > profiles/r03_bench_b32_kernel_stats_one_stream.md
python3 tools/profile_summary.py stats $O/two 10 "rocprofv3 --kernel-trace --stats summary, round 3: python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra (B=32, weight gradients on the second stream: the mode the bench value is measured in; overlapped kernels inflate each other's duration)" > profiles/r03_bench_b32_kernel_stats_two_streams.md
python3 tools/profile_summary.py stats $O/b4 12 "rocprofv3 --kernel-trace --stats summary, round 3: MD2_ONE_STREAM=1 python3 tools/prof_config.py 4 18 1 128 416 f32 10 (B=4, the reference's batch size; 12 steps in the trace)" > profiles/r03_b4_kernel_stats_one_stream.md
cp $O/one/*kernel_stats.csv profiles/r03_bench_b32_kernel_stats_one_stream.csv; cp $O/two/*kernel_stats.csv profiles/r03_bench_b32_kernel_stats_two_streams.csv; cp $O/b4/*kernel_stats.csv profiles/r03_b4_kernel_stats_one_stream.csv
python3 tools/profile_summary.py pmc $O/fetch $O/write 7 profiles/r03_pmc_hbm_traffic.json 32 1 128 416 18 f32 > profiles/r03_pmc_hbm_traffic.md
python3 tools/profile_summary.py valu $O/valu 7 profiles/r03_loss_pmc.json 32 1 128 416 > profiles/r03_loss_train_pmc.md
MD2_PMC_COMMAND="python3 tools/prof_config.py 8 18 3 320 1024 bf16 3" python3 tools/profile_summary.py pmc $O/fetch5 $O/write5 5 profiles/r03_cfg5_bf16_pmc_hbm_traffic.json 8 3 320 1024 18 bf16 > profiles/r03_cfg5_bf16_pmc_hbm_traffic.md
