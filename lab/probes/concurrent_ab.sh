for K in 8 16 32; do for b2 in default 0; do
  if [ $b2 = default ]; then unset MPST_B2; else export MPST_B2=0; fi
  python bench.py --steps 3 --warmup 2 --no-cpu-baseline --concurrent $K 2>/dev/null | tail -1 > gpurun_out/x_conc_${K}_$b2.json
done; done
