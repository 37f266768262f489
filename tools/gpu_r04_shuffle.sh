# round 4: north_star's "wavefront shuffles" settled by measurement -- N=1024, the last exchange as v_permlane32/16_swap + DPP
# instead of an LDS round trip: time (tools/tune.py, alternating, bits against the product) and package power (rocm-smi)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
S=$O/shuffle_ab.txt; : > $S
for rep in 1 2 3; do
  for lib in r4s_n9_prod r4s_n9_shuf; do
    echo "== $lib (rep $rep)" >> $S
    python tools/tune.py --streams 262144 --log-modes 1,0 --lib tools/bin/libglvtune_$lib.so --grids 0 --iters 10 --reps 5 2>/dev/null >> $S
  done
done
for ops in 256 6; do
  for lib in r4s_n9_prod r4s_n9_shuf; do
    echo "== $lib extra_ops=$ops (256: GL_R16 out; 6: fft+gravity+average F=5)" >> $S
    python tools/tune.py --streams 262144 --log-modes 1 --lib tools/bin/libglvtune_$lib.so --grids 0 --iters 10 --reps 5 --extra-ops $ops 2>/dev/null >> $S
  done
done
echo "== power: product" >> $S; python tools/power_probe.py --n 1024 --only f32 --seconds 5 2>/dev/null >> $S
echo "== power: shuffle" >> $S; GLV_SPECTRUM_LIB=$GRAFT_REPO_ROOT/glava_amd/csrc/libglvspectrum_shuf9.so python tools/power_probe.py --n 1024 --only f32 --seconds 5 2>/dev/null >> $S
echo "== power: product (again)" >> $S; python tools/power_probe.py --n 1024 --only f32 --seconds 5 2>/dev/null >> $S
echo "== power: shuffle (again)" >> $S; GLV_SPECTRUM_LIB=$GRAFT_REPO_ROOT/glava_amd/csrc/libglvspectrum_shuf9.so python tools/power_probe.py --n 1024 --only f32 --seconds 5 2>/dev/null >> $S
cat $S | cut -c1-220
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.txt | head -20
