#!/bin/bash
# Round-6 profile set, run on the GPU box from the repo root (tools/make_profiles.sh); raw output under gpurun_out/prof_final/, summaries by tools/profile_summary.py.
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
# matrix-pipe counters of every kernel (busy cycles, co-execution with vector instructions, MFMA count) and the LDS conflict counters, one pass each
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/lds -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/lds.log 2>&1
# ResNet-50 encoder, 640x192, batch 16 (BASELINE configs[3]): kernel stats, one stream
MD2_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r50 -o t -- python3 $R/tools/prof_config.py 16 50 3 192 640 f32 5 > $O/r50.log 2>&1
# configuration #5 (1024x320, bf16 activation storage = md2_config.precision 2, batch 8): kernel stats, and the HBM traffic of its kernel classes for the roofline block bench.py prints with that run
MD2_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5 -o t -- python3 $R/tools/prof_config.py 8 18 3 320 1024 bf16_storage 5 > $O/cfg5.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch5 -o t -- python3 $R/tools/prof_config.py 8 18 3 320 1024 bf16_storage 3 > $O/fetch5.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write5 -o t -- python3 $R/tools/prof_config.py 8 18 3 320 1024 bf16_storage 3 > $O/write5.log 2>&1
# vector-instruction counters of the C = 3 loss kernel at configuration #5 (its roofline entry, round 6), and the L1 -> L2 request counters of the headline step's kernels
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/valu5 -o t -- python3 $R/tools/prof_config.py 8 18 3 320 1024 bf16_storage 3 > $O/valu5.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/l2 -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/l2.log 2>&1 || echo "(L2 request counters not collected)"
cd $R
tail -1 $O/one.log | cut -c1-300; tail -1 $O/two.log | cut -c1-300; grep ms/step $O/b4.log
bash tools/summarize_profiles.sh
