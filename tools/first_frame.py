import sys, time
sys.path.insert(0, '/root/repo')
import bench
sc = bench.build_scene(12)
c = bench.make_caster(sc, 1920, 1080, 0, hit_records=0)
t0 = time.perf_counter(); assert c.compute(); t1 = time.perf_counter(); assert c.compute(); t2 = time.perf_counter()
print("first compute %.1f ms (code load + coarse table build), second %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print(c.memory_usage())
