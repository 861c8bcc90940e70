#!/usr/bin/env python3
"""Lab: samples the GPU's shader clock, memory clock and power from sysfs (hwmon of the amdgpu device; no privileges needed)
every `period` seconds into a CSV until it is terminated, for "does a VALU-bound kernel lose clock when it is sustained?"
(round 4: profiles/r04_half_clock.txt).  usage: python tools/clock_sampler.py out.csv [period=0.05]
   summary: python tools/clock_sampler.py --summary out.csv   (samples in the busiest half by power)"""
import glob
import os
import signal
import sys
import time


def sources():
    found = {}
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for key, names in (("sclk_hz", ["freq1_input"]), ("mclk_hz", ["freq2_input"]),
                           ("power_uw", ["power1_average", "power1_input"]), ("temp_mc", ["temp2_input", "temp1_input"])):
            for n in names:
                p = os.path.join(hw, n)
                if key not in found and os.path.exists(p):
                    found[key] = p
        if found:
            break
    return found


def read(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return -1


def summary(path):
    rows = [l.strip().split(",") for l in open(path) if l[0].isdigit()]
    rows = [(float(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in rows]
    if not rows:
        print("no samples")
        return
    rows.sort(key=lambda r: -r[3])
    busy = rows[:max(1, len(rows) // 2)]
    sclk = sorted(r[1] / 1e6 for r in busy)
    pw = sorted(r[3] / 1e6 for r in busy)
    print("samples %d (busiest half by power: %d)  sclk MHz min %.0f median %.0f max %.0f   power W min %.0f median %.0f max %.0f   mclk MHz %.0f"
          % (len(rows), len(busy), sclk[0], sclk[len(sclk) // 2], sclk[-1], pw[0], pw[len(pw) // 2], pw[-1], busy[0][2] / 1e6))


def main():
    if sys.argv[1] == "--summary":
        return summary(sys.argv[2])
    out, period = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    src = sources()
    stop = []
    signal.signal(signal.SIGTERM, lambda *a: stop.append(1))
    with open(out, "w") as f:
        f.write("# t_s,sclk_hz,mclk_hz,power_uw,temp_mc  sources: %s\n" % src)
        while not stop:
            f.write("%.4f,%d,%d,%d,%d\n" % (time.time(), read(src.get("sclk_hz", "")), read(src.get("mclk_hz", "")),
                                            read(src.get("power_uw", "")), read(src.get("temp_mc", ""))))
            f.flush()
            time.sleep(period)


if __name__ == "__main__":
    main()
