# depth-16 probe with an RSS watchdog (the box's cgroup allows ~300 GiB): bash tools/gpu_deep16.sh
mkdir -p gpurun_out
python tools/deep_scene_probe.py 16 ${1:-1920} ${2:-1080} ${3:-1} > gpurun_out/deep16.log 2>&1 &
PID=$!
PEAK=0
while kill -0 $PID 2>/dev/null; do
  RSS=$(awk '/VmRSS/ {print $2}' /proc/$PID/status 2>/dev/null || echo 0)
  [ "${RSS:-0}" -gt "$PEAK" ] && PEAK=$RSS
  if [ "${RSS:-0}" -gt 270000000 ]; then echo "RSS $RSS kB: killing" >> gpurun_out/deep16.log; kill -9 $PID; break; fi
  sleep 2
done
wait $PID 2>/dev/null
echo "peak RSS kB $PEAK" >> gpurun_out/deep16.log
tail -5 gpurun_out/deep16.log
