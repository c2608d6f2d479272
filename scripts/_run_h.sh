cd $GRAFT_REPO_ROOT
export CB_NOWGRAD=1
for rep in 1 2; do
for b in cb_base cb_cur cb_pre1; do
  echo "== $b plain rep$rep"; timeout 120 ./build/$b | grep fwd
  echo "== $b stats rep$rep"; CB_STATS=1 timeout 120 ./build/$b | grep fwd
done
done
for b in cb_base cb_cur cb_pre1; do echo "== $b add"; CB_ADD=1 timeout 120 ./build/$b | grep fwd; done
