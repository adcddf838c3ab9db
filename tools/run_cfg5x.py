import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as BC
r = BC.cfg5x(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
print({k: r[k] for k in r if k != "config"})
