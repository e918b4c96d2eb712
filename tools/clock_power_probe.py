"""Core clock and board power while the batched OO evaluation runs in a loop (tools only): a thread polls
rocm-smi while the main thread keeps the GPU busy; idle readings first."""
import os, sys, time, threading, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def poll(tag, seconds, out):
    t_end = time.time() + seconds
    while time.time() < t_end:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower", "-d", "0"],
                               capture_output=True, text=True, timeout=20).stdout
        except Exception as e:  # noqa: BLE001
            r = f"rocm-smi failed: {e}"
        keep = [l.strip() for l in r.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "power"))]
        out.append((tag, time.time(), keep))
        time.sleep(0.3)

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pqc, batch, single, thetas = bench.build_geometries([g % 16 for g in range(G)])
torch.cuda.synchronize()
log = []
poll("idle", 1.5, log)
th = threading.Thread(target=poll, args=("busy", 8.0, log))
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 9.0:
    for _ in range(50):
        batch.energy_and_gradient(thetas)
    torch.cuda.synchronize()
    n += 50
dt = time.time() - t0
th.join()
print(f"{n} batched calls of {G} geometries in {dt:.2f} s: {dt / n * 1e6:.1f} us per call")
for tag, t, keep in log[:2] + log[-6:]:
    print(tag, f"{t - t0:+.1f}s", " | ".join(keep))
