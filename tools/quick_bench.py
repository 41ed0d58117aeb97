"""quick bench: python tools/quick_bench.py N FRAMES [label] -> one short line (value, kernel_ms, frac)"""
import json, subprocess, sys
n, fr = sys.argv[1], sys.argv[2]
lab = sys.argv[3] if len(sys.argv) > 3 else ""
r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--frames", fr,
                    "--instances-per-gpu", n], capture_output=True, text=True)
try:
    j = json.loads(r.stdout.strip().splitlines()[-1])
    print(f"{lab:16s} N={n:>5s} frames={fr:>7s}  {j['value']/1000:8.1f} Gsamples/s  kernel {j['roofline']['kernel_ms']:8.3f} ms  frac {j['roofline']['frac']:.4f}")
except Exception as ex:
    print(lab, "FAILED", ex, r.stdout[-500:], r.stderr[-1500:])
