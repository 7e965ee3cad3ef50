for piv in 0 1; do
  if [ $piv = 1 ]; then export MPST_SS_PIVOT=1; else unset MPST_SS_PIVOT; fi
  echo "=== pivotwise=$piv"
  MPST_BIG_SYNC=1 MPST_SS_DEBUG=1 python lab/probes/subspace_rejects.py 2048 40 64 8 2 float64 3 2>&1 | grep -v "chol phases" | grep "^sweep\|st 0 1" | cut -c1-200 | tail -22
done
