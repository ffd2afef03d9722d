#!/bin/bash
# ON THE GPU BOX: wgrad kernel times for library variants. usage: bash tools/libsweep.sh <lib|default> ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  export SW_LIB_PATH=$REPO/$lib
  [ "$lib" = "default" ] && unset SW_LIB_PATH
  echo "== $lib"
  bash $REPO/tools/kstats.sh ls_$(basename $lib .so) 0 2>&1 | grep -E "wgrad_partial|total"
done
