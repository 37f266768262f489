#!/bin/bash
# tools/chain_var.sh [libsuffix ...] -- same-box A/B of product library variants (glava_amd/csrc/libglvspectrum<suffix>.so; "" = the product), alternating processes:
# wall ms per call of tools/cfg_run.py for the stateful chains.  The stateful chains repeat to ~0.5 % within a process but differ by up to 10 % from process to
# process and box to box (profiles/r05/run_to_run.txt): a change is only believed from alternating runs on ONE box.
cd "$GRAFT_REPO_ROOT" || exit 1
ms() { python tools/cfg_run.py $1 80 2>&1 | tail -1 | sed 's/.*wall ms per call \([0-9.]*\).*/\1/'; }
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=("")
for i in 1 2 3; do
 for lib in "${LIBS[@]}"; do
  export GLV_SPECTRUM_LIB=$GRAFT_REPO_ROOT/glava_amd/csrc/libglvspectrum$lib.so
  echo "lib '$lib': chain $(ms chain)  gl_default $(ms gl_default)  gl_bars $(ms gl_bars)  gl_sm64 $(ms gl_sm64)  gl_sm $(ms gl_sm)  configs2 $(ms configs2)  n1024bars $(ms n1024bars)"
 done
done
