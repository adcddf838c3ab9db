import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as BC
print(BC.cfg3(int(sys.argv[1]) if len(sys.argv) > 1 else 3))
