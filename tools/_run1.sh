mkdir -p gpurun_out
for s in "" 16 "16,4" "16,4,4,2" "8" "8,4" "4" "16,2" "16,8,4"; do
  echo "== sched '$s'"
  PRAG_MM_GROWTH_SCHED=$s C3_ONLY16=1 C3_REPS=30 python tools/c3_search.py 2>&1 | grep "shadow=0"
done
