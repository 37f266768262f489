cd $GRAFT_REPO_ROOT
for i in 1 2; do
 for lib in "" _nofence; do
  export GLV_SPECTRUM_LIB=$GRAFT_REPO_ROOT/glava_amd/csrc/libglvspectrum$lib.so
  echo "== lib '$lib'"; python tools/cfg_run.py chain 100 2>&1 | tail -1 | cut -c80-200; python tools/cfg_run.py gl_default 100 2>&1 | tail -1 | cut -c80-200; python tools/cfg_run.py gl_bars 100 2>&1 | tail -1 | cut -c80-200;  python tools/cfg_run.py configs2 100 2>&1 | tail -1 | cut -c80-200
 done
done
