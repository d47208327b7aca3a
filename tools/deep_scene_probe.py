import sys, os, time, json
sys.path.insert(0, "/root/repo")
import bench, numpy as np
t=time.time(); D = int(sys.argv[1]) if len(sys.argv) > 1 else 14
t=time.time(); sc = bench.build_scene(D); print("build s", round(time.time()-t,1), "descriptors", sc["octree"].descriptor_buffer.size, flush=True)
c = bench.make_caster(sc, 1920, 1080, 0)
for _ in range(2): assert c.compute(), c.last_error()
c.timing_reset()
for _ in range(5): assert c.compute()
n, ms = c.timing(); ctr = c.counters()
rays = ctr["primary_rays"] + ctr["shadow_rays"]
print(json.dumps({"depth": D, "kernel_ms": round(ms/n,3), "Mrays/s": round(rays/(ms/n)/1e3,1), "steps": ctr["steps"], "descriptor_reads": ctr["descriptor_reads"]}))
