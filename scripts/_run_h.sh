cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in cb_base cb_wr1 cb_wp; do
  echo "== $b plain rep$rep"; timeout 200 ./build/$b | sed 's/wgrad/fwd  /' | awk 'NR%2==0'
done
done
