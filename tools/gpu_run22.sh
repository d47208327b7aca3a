mkdir -p gpurun_out
python - > gpurun_out/r22.log 2>&1 <<'PY'
import sys, json; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import bench
sc = bench.build_scene(12)
c = bench.make_caster(sc, 1920, 1080, 0)
c.add_to_settings_buffer("jump_min_run", "J", 1 << 24)
c.add_to_settings_buffer("event_threshold", "E", 64)
c.add_to_settings_buffer("shade_threshold", "S", 64)
for name, vals in (("event_threshold", [16, 32, 48, 64, 96, 128, 256, 1024]), ("shade_threshold", [1, 16, 32, 48, 64])):
    for v in vals:
        c.overwrite_setting(name, v)
        for _ in range(2): assert c.compute()
        c.timing_reset()
        for _ in range(5): assert c.compute()
        n, ms = c.timing()
        print(json.dumps({name: v, "kernel_ms": round(ms / n, 3)}), flush=True)
    c.overwrite_setting(name, 64)
PY
