cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-trace}; do
  export FASTNERF_LIB=$PWD/fast-learning-nerf_amd/variants/$v.so
  echo "== $v"; python tools/time_dw16.py | tail -1
  for m in save inf; do
  BF_ONE_WG=1 python tools/trace_fwd.py $m | grep "shader clock\|whole tile"
  python tools/trace_fwd.py $m | grep "shader clock\|whole tile"
  done
done
