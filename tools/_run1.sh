for i in 1 2 3; do for q in 0 999999999; do
  echo "quad_min_rows=$q: $(PRAG_SCAN8_QUAD_ROWS=$q python tools/shard_pass.py | cut -c1-26)"
done; done
