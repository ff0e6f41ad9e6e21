cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 python3 $R/bench.py --steps 20 --warmup 3 > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err; tail -c 600 $R/gpurun_out/bench_default.json
timeout 600 python3 $R/bench.py --steps 20 --warmup 3 --queries 1000 --docs 1000000 --no-cpu-baseline > $R/gpurun_out/bench_c3.json 2> $R/gpurun_out/bench_c3.err; cat $R/gpurun_out/bench_c3.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'], d['recall_at_k_vs_oracle'], d['topk_ids_bit_exact_vs_oracle'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3e -- python3 $R/tools/c3_search.py 2>&1 | tail -1
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $C | cut -d' ' -f1)
  C3_REPS=2 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c3_$n -- python3 $R/tools/c3_search.py > $R/gpurun_out/pmc_c3_$n.log 2>&1
  tail -1 $R/gpurun_out/pmc_c3_$n.log
done
