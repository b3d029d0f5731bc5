import sys, json
sys.path.insert(0, 'tools')
import bench_polar as b
for want, precise in ((("xolp",), False), (("xolp","normals"), False)):
    print(json.dumps(b.time_variant(128, want, realistic=True, precise=precise)), flush=True)
