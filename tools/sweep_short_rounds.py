import sys, json
sys.path.insert(0, '.')
import bench
sc = bench.build_scene(12)
c = bench.make_caster(sc, 1920, 1080, 0, hit_records=0)
def t(c, n=10):
    for _ in range(2): assert c.compute()
    c.timing_reset()
    for _ in range(n): assert c.compute()
    k, ms = c.timing(); return ms / k
print("default (tuned instance)", round(t(c), 4))
assert c.add_to_settings_buffer("safe_steps", "SAFE_STEPS", 16)
assert c.add_to_settings_buffer("jump_min_run", "JUMP_MIN_RUN", 16)
assert c.add_to_settings_buffer("exact_steps", "EXACT_STEPS", 16)
for ss in (8, 16, 24):
    for k in (8, 10, 12, 16, 20, 24):
        c.overwrite_setting("safe_steps", ss); c.overwrite_setting("jump_min_run", k)
        print("safe_steps", ss, "jump_min_run", k, round(t(c), 4), flush=True)
