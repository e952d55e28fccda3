"""CBIR leg only: both search methods at BASELINE.json's size, bit-equality between them, per-kernel time split.
usage: python tools/cbir_bench.py [nq] [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
out = bench.bench_cbir(torch.device("cuda:0"), nq=nq, n=n, with_cpu=False)
print(json.dumps(out))
