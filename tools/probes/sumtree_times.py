import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "i-dqn_amd")]
import bench, json
print(json.dumps({k: v for k, v in bench.sampling_leg().items() if k.startswith("sumtree")}, indent=0))
