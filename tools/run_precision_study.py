import sys, json; sys.path.insert(0,'.')
import bench
print(json.dumps(bench.precision_study(), indent=1))
