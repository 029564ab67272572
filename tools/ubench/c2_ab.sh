for rep in 1 2; do
for mode in full sweep none; do
  if [ $mode = full ]; then export JRLQP_DENSE_SPECULATE=1; elif [ $mode = sweep ]; then export JRLQP_DENSE_SPECULATE=sweep; else export JRLQP_DENSE_SPECULATE=0; fi
  echo -n "$mode: "; timeout -k 10 120 python tools/time_dense.py 2>&1 | grep "^GPU"
done; done
