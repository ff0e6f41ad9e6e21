cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest $R/tests/test_gpu_index.py -x -q -k "3000-4200-5-256-1-f32" 2>&1 | grep -E "Error|error|assert|Mismatch|mismatch|Max|FAILED|passed" | head -12
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum FETCH_SIZE"; do
  i=$((i+1))
  C3_REPS=2 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mm_$i -- python3 $R/tools/c3_search.py > $R/gpurun_out/pmc_mm_$i.log 2>&1
  tail -1 $R/gpurun_out/pmc_mm_$i.log
done
