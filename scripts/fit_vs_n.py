"""fits/s of bench.py's workload (3-D SE + noise, inputs resident in HBM) at several N: one line per size."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for n in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384, 32768]:
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--n", str(n), "--steps", "10", "--warmup", "3",
                          "--no-cpu-baseline", "--no-predict", "--no-configs"], capture_output=True, text=True).stdout
    d = json.loads([ln for ln in out.splitlines() if ln.startswith('{"metric"')][-1])
    t = d["ms_per_step"]
    print(f"N={n:6d}: {d['value']:9.2f} fits/s  {t:9.3f} ms per fit  {n ** 3 / 3 / t / 1e9:6.2f} TFLOP/s overall (n^3/3 flop)  "
          f"self-check {d['self_check']['max_rel_residual']:.1e}", flush=True)
