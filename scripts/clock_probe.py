"""Shader clock and board power, sampled from sysfs by a thread (~0.5 ms period), while (1) idle, (2) the graded kernel alone,
(3) the hot-path step, (4) the backward kernel alone run in loops -- is the step on a slow box of the pool clocked lower?"""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flownet2-pytorch_amd"))
import torch
import bench

HW = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
print("hwmon dirs:", len(HW), flush=True)
def rd(f):
    try:
        with open(f) as h:
            return float(h.read().split()[0])
    except Exception:
        return float("nan")
samples, phase, stop = [], ["idle"], [False]
def sampler():   # every card the node exposes: the one whose power follows our phases is ours
    while not stop[0]:
        samples.append((phase[0], [(rd(d + "/freq1_input") / 1e6, rd(d + "/power1_input") / 1e6) for d in HW]))
        time.sleep(0.002)
dev = torch.device("cuda:0")
hp = bench.HotPath(dev, 1234)
for _ in range(5): hp.step()
torch.cuda.synchronize()
th = threading.Thread(target=sampler, daemon=True); th.start()
def run(name, fn, n):
    torch.cuda.synchronize(); phase[0] = name
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e6
    phase[0] = "idle"; time.sleep(0.05)
    return dt
time.sleep(0.2)
res = {}
for rep in range(2):
    res["corr_fwd alone"] = run("fwd", hp.corr_fwd, 4000)
    res["step"] = run("step", hp.step, 1500)
    res["corr_bwd alone"] = run("bwd", hp.corr_bwd, 2000)
    res["step again"] = run("step2", hp.step, 1500)
stop[0] = True; th.join()
import statistics as st
ours = max(range(len(HW)), key=lambda ci: max(smp[1][ci][1] for smp in samples))   # the card whose power peaks highest
print("our card:", HW[ours].split("/")[4], "power cap W", rd(HW[ours] + "/power1_cap") / 1e6, "temps C", [rd(f) / 1e3 for f in sorted(glob.glob(HW[ours] + "/temp*_input"))])
for ci, d in enumerate(HW):
    if ci != ours: continue
    line = d.split("/")[4] + ":"
    for ph in ("idle", "fwd", "step", "bwd", "step2"):
        x = [smp[1][ci] for smp in samples if smp[0] == ph]
        if x:
            line += "  %s sclk %.0f/%.0f/%.0f W %.0f/%.0f/%.0f" % (ph, min(v[0] for v in x), st.median(v[0] for v in x), max(v[0] for v in x), min(v[1] for v in x), st.median(v[1] for v in x), max(v[1] for v in x))
    print(line)
print({k: round(v, 1) for k, v in res.items()}, "us per call")
