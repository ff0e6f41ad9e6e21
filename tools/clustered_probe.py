import sys, json
sys.path.insert(0, "/root/repo")
import torch, bench
import probing_rag_amd as pra
print(json.dumps(bench.clustered_variant(torch, pra, 768, 10), indent=1))
