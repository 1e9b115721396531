#!/bin/bash
# Turns the raw rocprofv3 output of tools/make_profiles.sh (gpurun_out/prof_final/, merged back from the GPU box) into the summaries kept under profiles/.
# Run from the repo root, on any machine: files written under profiles/ on the GPU box do not travel back, gpurun_out/ does.
R=$PWD; O=$R/gpurun_out/prof_final
python3 tools/profile_summary.py stats $O/one 10 "rocprofv3 --kernel-trace --stats summary, round 6: MD2_ONE_STREAM=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra (B=32; 10 steps in the trace: 2 warm-up + 5 timed + 3 profiled). Every kernel alone: the mode the bench roofline is timed in." > profiles/r06_bench_b32_kernel_stats_one_stream.md
python3 tools/profile_summary.py stats $O/two 10 "rocprofv3 --kernel-trace --stats summary, round 6: python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra (B=32, weight gradients on the second stream: the mode the bench value is measured in; overlapped kernels inflate each other's duration)" > profiles/r06_bench_b32_kernel_stats_two_streams.md
python3 tools/profile_summary.py stats $O/b4 12 "rocprofv3 --kernel-trace --stats summary, round 6: MD2_ONE_STREAM=1 python3 tools/prof_config.py 4 18 1 128 416 f32 10 (B=4, the reference's batch size; 12 steps in the trace)" > profiles/r06_b4_kernel_stats_one_stream.md
cp $O/one/*kernel_stats.csv profiles/r06_bench_b32_kernel_stats_one_stream.csv; cp $O/two/*kernel_stats.csv profiles/r06_bench_b32_kernel_stats_two_streams.csv; cp $O/b4/*kernel_stats.csv profiles/r06_b4_kernel_stats_one_stream.csv
python3 tools/profile_summary.py pmc $O/fetch $O/write 7 profiles/r06_pmc_hbm_traffic.json 32 1 128 416 18 f32 > profiles/r06_pmc_hbm_traffic.md
python3 tools/profile_summary.py valu $O/valu 7 profiles/r06_loss_pmc.json 32 1 128 416 > profiles/r06_loss_train_pmc.md
MD2_PMC_COMMAND="python3 tools/prof_config.py 8 18 3 320 1024 bf16_storage 3" python3 tools/profile_summary.py pmc $O/fetch5 $O/write5 5 profiles/r06_cfg5_bf16_storage_pmc_hbm_traffic.json 8 3 320 1024 18 bf16_storage > profiles/r06_cfg5_bf16_storage_pmc_hbm_traffic.md
python3 tools/profile_summary.py stats $O/cfg5 7 "rocprofv3 --kernel-trace --stats summary, round 6: MD2_ONE_STREAM=1 python3 tools/prof_config.py 8 18 3 320 1024 bf16_storage 5 (BASELINE configs[4]: 1024x320, batch 8, bf16 activation storage = md2_config.precision 2; 7 steps in the trace)" > profiles/r06_cfg5_bf16_storage_kernel_stats_one_stream.md
cp $O/cfg5/*kernel_stats.csv profiles/r06_cfg5_bf16_storage_kernel_stats_one_stream.csv
python3 tools/profile_summary.py stats $O/r50 7 "rocprofv3 --kernel-trace --stats summary, round 6: MD2_ONE_STREAM=1 python3 tools/prof_config.py 16 50 3 192 640 f32 5 (ResNet-50 encoder, 640x192, batch 16: BASELINE configs[3]; 7 steps in the trace)" > profiles/r06_r50_kernel_stats_one_stream.md
cp $O/r50/*kernel_stats.csv profiles/r06_r50_kernel_stats_one_stream.csv
python3 tools/profile_summary.py mfma $O/mfma $O/lds $O/valu profiles/r06_mfma_pmc.json > profiles/r06_mfma_pmc.md
python3 tools/profile_summary.py valu $O/valu5 5 profiles/r06_cfg5_loss_pmc.json 8 3 320 1024 > profiles/r06_cfg5_loss_train_pmc.md
[ -d $O/l2 ] && python3 tools/profile_summary.py l2 $O/l2 7 > profiles/r06_l2_requests.md
