for dbg in 0 1 3 7; do
  MPST_IMB_DBG=$dbg python bench.py --workload impute --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/x_imp_d$dbg.json 2>/dev/null
done
MPST_IMP_NO_BATCH=1 python bench.py --workload impute --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/x_imp_o.json 2>/dev/null
