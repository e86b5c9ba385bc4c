"""Round 6: the same process steps at ~11.7 ms ("fast") or ~14.7 ms ("slow") and can change from one to the other; in the slow state the
HBM-bound work (optimizer epilogues, LayerNorm / attention backward) takes ~2x as long, MFMA-bound kernels the same.  This monitor steps
the bench workload for `seconds` and prints, per window of 10 steps: ms/step, and what sysfs says about the device at that moment
(hwmon power / clocks / temperatures, pp_dpm_* current levels, gpu_busy / mem_busy percent).  No child processes, nothing written.
usage: python tools/mode_monitor.py [seconds] [config] [batch]"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import afft_amd
from afft_amd.parallel import Trainer

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
afft_amd.set_precision("bf16")
dev = torch.device("cuda:0")
pr = torch.cuda.get_device_properties(0)
bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
devdir = next((os.path.join(c, "device") for c in sorted(glob.glob("/sys/class/drm/card[0-9]*"))
               if os.path.basename(os.path.realpath(os.path.join(c, "device"))).lower() == bdf), None)
files = {}
if devdir:
    for hw in glob.glob(os.path.join(devdir, "hwmon", "hwmon*")):
        for f in sorted(os.listdir(hw)):
            if f.endswith("_input") or f.endswith("_average"):
                lab = os.path.join(hw, f.rsplit("_", 1)[0] + "_label")
                files[(open(lab).read().strip() if os.path.exists(lab) else "") + ":" + f] = os.path.join(hw, f)
    for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent", "mem_busy_percent", "power_dpm_force_performance_level"):
        if os.path.exists(os.path.join(devdir, f)):
            files[f] = os.path.join(devdir, f)


def snap():
    out = {}
    for k, p in files.items():
        try:
            t = open(p).read().strip()
        except Exception as ex:  # noqa: BLE001
            t = "?" + type(ex).__name__
        if k.startswith("pp_dpm"):
            cur = [ln for ln in t.splitlines() if "*" in ln]
            t = (cur[0] if cur else t.replace("\n", "|"))[:24]
        out[k] = t
    return out


print("device", bdf, "sysfs", devdir, "files", sorted(files))
model, c = B.build_model(name, dev)
feats, tgt, sub = B.make_inputs(c, batch, c["T"], 0, dev)
tr = Trainer(model, {"cls_action": 1.0, "past_cls_action": 1.0, "past_reg": 1.0})
model.train()
for _ in range(8):
    tr.step(feats, tgt, sub)
torch.cuda.synchronize()
t_end = time.perf_counter() + seconds
w = 0
while time.perf_counter() < t_end:
    t0 = time.perf_counter()
    for i in range(10):
        tr.step(feats, tgt, sub)
        if i == 5:
            s = snap()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"w{w:03d} t={time.perf_counter() - (t_end - seconds):6.1f}s {ms:6.2f} ms/step | " + " ".join(f"{k}={v}" for k, v in s.items()), flush=True)
    w += 1
