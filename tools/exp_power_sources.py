"""Which telemetry source tells the truth about board power / shader clock under load?  Lists the amdgpu hwmon files and reads
them next to rocm-smi while the inference forward runs back to back (bench.py's PowerSampler picks its source from this)."""
import glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
    name = open(os.path.join(hw, "name")).read().strip() if os.path.exists(os.path.join(hw, "name")) else "?"
    print(hw, name, sorted(f for f in os.listdir(hw) if f.startswith(("power", "freq"))))
    for f in sorted(os.listdir(hw)):
        if f.endswith("_label"):
            print("   ", f, open(os.path.join(hw, f)).read().strip())
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
def load(sec):
    t0 = time.time()
    while time.time() - t0 < sec:
        for _ in range(20):
            a @ a
        torch.cuda.synchronize()
def smi():
    return subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()[1:2]
for phase in ("idle", "load"):
    th = threading.Thread(target=load, args=(4.0,)) if phase == "load" else None
    if th: th.start()
    time.sleep(1.0)
    for i in range(4):
        row = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            for f in ("power1_average", "power1_input", "freq1_input", "freq2_input"):
                p = os.path.join(hw, f)
                if os.path.exists(p):
                    try: row.append(f"{f}={open(p).read().strip()}")
                    except Exception as e: row.append(f"{f}=ERR({e})")
        print(phase, i, " ".join(row), "| smi:", smi())
        time.sleep(0.5)
    if th: th.join()
with bench.PowerSampler() as ps:
    load(3.0)
print("PowerSampler under load:", ps.source, ps.summary())
