"""Board power and shader clock while the FP8 GEMM runs back to back (evidence for the power-limited reading of the 50 % figure).
Samples sysfs hwmon (power1_average / power1_input, freq1_input) and `rocm-smi` if available; random vs all-zero operands."""
import glob, json, os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_fp8_amd.pytorch import ops

dev = torch.device("cuda:0")


def read_hwmon():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "freq1_input", "power1_cap"):
            p = os.path.join(h, name)
            if os.path.exists(p):
                try:
                    out[f"{h.split('/')[4]}:{name}"] = int(open(p).read().strip())
                except Exception as e:  # noqa
                    out[f"{h.split('/')[4]}:{name}"] = str(e)
    return out


def read_smi():
    try:
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        return json.loads(r.stdout) if r.stdout.strip().startswith("{") else r.stdout[-400:] + r.stderr[-400:]
    except Exception as e:  # noqa
        return str(e)


print("idle hwmon:", read_hwmon())
print("idle smi:", json.dumps(read_smi())[:1500])
M, N, K = 8192, 8192, 8192
one = torch.ones(1, device=dev)
for label in ("random", "zeros"):
    if label == "random":
        a = torch.randint(0, 256, (M, K), device=dev, dtype=torch.uint8); b = torch.randint(0, 256, (N, K), device=dev, dtype=torch.uint8)
        for t in (a, b):
            t[(t & 0x7F) >= 0x78] &= 0x3F
    else:
        a = torch.zeros((M, K), device=dev, dtype=torch.uint8); b = torch.zeros((N, K), device=dev, dtype=torch.uint8)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    samples, stop = [], False

    def sampler():
        while not stop:
            samples.append((time.time(), read_hwmon()))
            time.sleep(0.2)

    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 6.0:
        for _ in range(50):
            ops.gemm_fp8(a, b, one, one, 0, 0, out=out, algo=4)
        torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    smi = read_smi()
    stop = True; th.join()
    print(f"{label}: {2.0 * M * N * K * n / dt / 1e12:.0f} TFLOP/s over {dt:.1f} s")
    keys = sorted({k for _, s in samples for k in s})
    for k in keys:
        vals = [s[k] for _, s in samples[3:] if isinstance(s.get(k), int)]
        if vals:
            print(f"   {k}: mean {sum(vals) / len(vals):.0f}  min {min(vals)}  max {max(vals)}  (n={len(vals)})")
    print("   smi under load:", json.dumps(smi)[:1200])

# ---- the whole training step (3B, b16 x s512): average board power / clock over ~8 s of steps
if "--step" in sys.argv:
    from llm_fp8_amd import train
    cfg = train.TrainingConfig(model_name="llama-3.2-3b", batch_size=16, max_seq_length=512, mixed_precision="fp8", use_te=True)
    torch.manual_seed(0)
    model = train.prepare_model(train.create_model(cfg, dev), cfg)
    opt, sched = train.create_optimizer(model, cfg)
    model.train()
    batch = train.synthetic_batch(cfg, model.config.vocab_size, dev)
    for _ in range(3):
        train.train_step(model, batch, opt, sched, cfg)
    torch.cuda.synchronize()
    samples, stop = [], False

    def sampler2():
        while not stop:
            samples.append(read_hwmon())
            time.sleep(0.05)

    th = threading.Thread(target=sampler2); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 8.0:
        train.train_step(model, batch, opt, sched, cfg); n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize(); dt = time.time() - t0
    stop = True; th.join()
    print(f"training step: {n / dt * 16 * 512:.0f} tokens/s over {dt:.1f} s ({1e3 * dt / n:.1f} ms/step)")
    cards = sorted({k.split(':')[0] for s in samples for k in s})
    best = max(cards, key=lambda c: sum(s.get(c + ':power1_input', 0) for s in samples))
    pw = [s[best + ':power1_input'] / 1e6 for s in samples if best + ':power1_input' in s]
    fq = [s[best + ':freq1_input'] / 1e6 for s in samples if best + ':freq1_input' in s]
    print(f"   {best}: power mean {sum(pw) / len(pw):.0f} W (min {min(pw):.0f}, max {max(pw):.0f}); sclk mean {sum(fq) / len(fq):.0f} MHz (min {min(fq):.0f}, max {max(fq):.0f}); n={len(pw)}")
