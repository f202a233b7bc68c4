"""Dev tool (GPU box): how often does a fresh process die in torch.distributed's start-up ('std::bad_variant_access' seen once in ~35 `bench.py --mode tp` runs of round 6)?
usage: python3 tools/init_stress.py <runs> [ENV=VAL ...]   — every run is a fresh child: init_process_group(nccl, device_id) at world 1, one all_reduce, destroy."""
import os, subprocess, sys, time
CHILD = r'''
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561"); os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
mode = os.environ.get("INIT_MODE", "device_id")
if mode == "device_id": dist.init_process_group("nccl", device_id=dev)
elif mode == "lazy": dist.init_process_group("nccl")
else: dist.init_process_group("gloo")
t = torch.ones(4, device=dev if mode != "gloo" else "cpu"); dist.all_reduce(t); dist.barrier(); dist.destroy_process_group(); print("ok")
'''
runs = int(sys.argv[1]); env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
for kv in sys.argv[2:]:
    k, v = kv.split("="); env[k] = v
bad, t0, msgs = 0, time.time(), {}
for i in range(runs):
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 or "ok" not in r.stdout:
        bad += 1
        key = (r.stderr.strip().splitlines() or ["?"])[-1][:160]
        msgs[key] = msgs.get(key, 0) + 1
print(f"{' '.join(sys.argv[2:]) or 'default'}: {runs} runs, {bad} failed, {time.time() - t0:.0f} s", msgs, flush=True)
