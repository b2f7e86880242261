cd $GRAFT_REPO_ROOT
for d in 1 4 5; do
  EXTRA_OPTS=kernel_dbg=$d bash tools/valu_share.sh valu_gen_dbg$d gen
done
