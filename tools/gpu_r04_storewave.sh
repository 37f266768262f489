# round 4, VERDICT r3 item 2: store-wave specialisation A/B at N=4096 x 65536 streams (f32 and GL_R16 output), one call, alternating
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
S=$O/storewave_ab.txt; : > $S
for rep in 1 2; do
  for ops in 0 256; do
    for lib in r4w_n11_base r4w_n11_sw; do
      echo "== $lib extra_ops=$ops (0: f32 spectra out, 256: GL_R16 texels out) rep $rep" >> $S
      timeout 300 python tools/tune.py --streams 65536 --log-modes 1 --lib tools/bin/libglvtune_$lib.so --grids 256,512,1024 --iters 10 --reps 3 --extra-ops $ops 2>&1 | grep -v "^$" | tail -12 >> $S
    done
  done
done
cat $S | cut -c1-230
