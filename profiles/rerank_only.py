"""Only bench.py's rerank leg (two cross-encoder shapes, 32 pairs): python profiles/rerank_only.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
class A: pass
print(json.dumps(bench.rerank_leg(A(), 0), indent=1))
