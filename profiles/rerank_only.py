import sys, json
sys.path.insert(0, '.')
import bench
class A: pass
print(json.dumps(bench.rerank_leg(A(), 0), indent=1))
