for v in evA evA2 evB evC; do
  echo "== $v"; NANOCALLER_HIP_LIB=$GRAFT_REPO_ROOT/build_exp/libnc_$v.so TOP=40 tools/prof_quick.sh $v python tools/bench_indel_pipe.py 64444167 2 2>&1 | grep -E "k_event_tiles|k_pick"
done
