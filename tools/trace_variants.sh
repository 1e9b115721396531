export TMPDIR=/tmp
cp monodepth2.jl_amd/libmd2hip.so /tmp/libmd2hip_tree.so; cp tools/ab/libmd2hip_tune.so monodepth2.jl_amd/libmd2hip.so
i=0
for v in "X=0" "MD2_NO_HEADS_ASIDE=1" "MD2_NO_DEFER_TAIL=1" "MD2_NO_HEADS_ASIDE=1 MD2_NO_DEFER_TAIL=1"; do
  i=$((i+1)); export $v
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tv$i -o t -- python3 tools/prof_config.py 4 18 1 128 416 f32 8 > gpurun_out/tv$i.log 2>&1
  echo "== $v: $(grep ms/step gpurun_out/tv$i.log)"; python3 tools/stream_trace.py $(find gpurun_out/tv$i -name "*kernel_trace.csv" | head -1) 2>&1 | head -9
  for n in $v; do unset ${n%%=*}; done
done
cp /tmp/libmd2hip_tree.so monodepth2.jl_amd/libmd2hip.so
