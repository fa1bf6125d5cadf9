#!/bin/bash
# SQ counters of wg_hash_k under variants of libtbk.so
cd "$GRAFT_REPO_ROOT"
cp tiebrush_amd/_build/libtbk.so /tmp/libtbk_orig.so
for e in "$@"; do
  if [ $e = cur ]; then cp /tmp/libtbk_orig.so tiebrush_amd/_build/libtbk.so; else cp gpurun_exp/$e/libtbk.so tiebrush_amd/_build/libtbk.so; fi
  echo "== $e"
  PMC_OUT=gpurun_out/wgpmc_$e bash tools/pmc_sq.sh p wg_hash_k
done
cp /tmp/libtbk_orig.so tiebrush_amd/_build/libtbk.so
