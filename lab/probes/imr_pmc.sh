cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for p in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $p | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc $p -d /tmp/pmc_$tag -o x --output-format csv -- python3 $R/bench.py --workload impute --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
  python3 $R/lab/probes/pmc_sum.py /tmp/pmc_$tag "k_imp_right<float, true"
done
