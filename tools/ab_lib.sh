#!/bin/bash
# A/B of several builds of the library on one box: tools/ab_lib.sh "<a.so> <b.so> ..." <command...>  -> runs the command once per library
# (copied over the in-tree one, "tree" = the in-tree build itself), restores the in-tree one; outputs gpurun_out/ab_<n>.log
libs=$1; shift
mkdir -p gpurun_out
cp -f transformer_tts_amd/libfs2_hip.so /tmp/fs2_tree.so
n=0
for l in $libs; do
  if [ "$l" = tree ]; then cp -f /tmp/fs2_tree.so transformer_tts_amd/libfs2_hip.so; else cp -f "$l" transformer_tts_amd/libfs2_hip.so; fi
  timeout -k 10 400 "$@" > gpurun_out/ab_$n.log 2>&1
  echo "ab_$n ($l) rc=$?"
  n=$((n+1))
done
cp -f /tmp/fs2_tree.so transformer_tts_amd/libfs2_hip.so
