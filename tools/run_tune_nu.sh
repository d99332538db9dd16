for model in mantle block; do
  for n in 2049 1025; do
    for nu in 3,3 2,2 2,3 2,1 1,2 1,1; do
      python tools/tune_nu.py $model $n 1,1 $nu 2>/dev/null | tail -1
    done
  done
done
