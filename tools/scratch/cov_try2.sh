#!/bin/bash
set -e
for cfg in "" "TBK_SCAN=lookback" "TBK_PREP_BLOCKS=4096" "TBK_PREP_BLOCKS=16384"; do
  echo "== $cfg"
  ( [ -n "$cfg" ] && export $cfg; timeout -k 10 200 python tools/cov_prof.py c3 64 5000000 8 2>/dev/null | sed -n '2,16p' )
done
