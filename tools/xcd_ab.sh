# same-box A/B of the XCD-aware workgroup order of conv_mfma32_kernel ("xcd_order" bits: 1 linears, 2 stride-2 convs, 4 the rest)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for o in 0 1 2 3; do
  echo -n "[enc xcd_order=$o] "; DISSC_OPTIONS=xcd_order=$o python tools/encode_bench.py --iters 10 2>/dev/null | tail -1
done; done
for o in 0 3; do
  echo -n "[enc hubert_split=0 xcd_order=$o] "; DISSC_OPTIONS=hubert_split=0,xcd_order=$o python tools/encode_bench.py --iters 10 2>/dev/null | tail -1
done
REPS=2 bash tools/opt_ab.sh "xcd_order=0" "xcd_order=4"
