import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import bench
args = bench.parse_args(["--steps","50","--warmup","5","--no-cpu-baseline"])
g = bench.geometry("4k", 0, 32)
r = bench.GpuRunner(args, g, 0, 0)
for _ in range(5): r.step()
r.sync()
for steps in (10, 50, 200, 800):
    t0=time.perf_counter(); r.mark(0)
    for _ in range(steps): r.step()
    t_issue=time.perf_counter()-t0
    r.mark(1); r.sync(); t_all=time.perf_counter()-t0
    ev=r.event_ms()
    print("steps %4d: cpu issue %.1f us/launch, wall %.1f us/launch, events %.1f us/launch" % (steps, t_issue/steps/2*1e6, t_all/steps/2*1e6, ev/steps/2*1e3))
