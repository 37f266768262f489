# round 4: A/B of the GL_R16 epilogue's knobs at N=4096 (one gpurun call, alternating, twice) + the fixed tests
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
timeout 900 python -m pytest tests/test_gl_fused.py tests/test_stream_order.py tests/test_gpu_parity.py tests/test_multi.py -q -m gpu > $O/pytest_new2.txt 2>&1
tail -4 $O/pytest_new2.txt | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/pytest_new2.txt | head
A=$O/ab_gl16.txt; : > $A
for rep in 1 2; do
  echo "== product (div_frames, block = lane, two frames per trip)" >> $A; python tools/gl_bench.py 2>/dev/null >> $A
  for v in gl_div0 gl_blk8 gl_blk4 gl_pair0; do
    echo "== $v" >> $A; GLV_SPECTRUM_LIB=$GRAFT_REPO_ROOT/glava_amd/csrc/libglvspectrum_$v.so python tools/gl_bench.py 2>/dev/null >> $A
  done
done
cat $A
