#!/bin/bash
# ON THE GPU BOX: disc_bwd / wgrad kernel times with riders on, for library variants. usage: bash tools/ridesweep.sh <lib> [<lib> ...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  export SW_LIB_PATH=$REPO/$lib
  [ "$lib" = "default" ] && unset SW_LIB_PATH
  echo "== $lib"
  SW_COSCHED=1 bash $REPO/tools/kstats.sh rs_$(basename $lib .so) 0 2>&1 | grep -E "disc_bwd|dec_rollout_bwd|enc_lstm_bwd|wgrad_partial|total"
done
