"""Why is step_adjoint_kernel bimodal between processes (2.65 vs 2.88 ms at 512^3)?  One fresh process per
call: runs the bench workload, prints the per-stage times next to the virtual addresses of the arrays the adjoint kernel
touches, so that fast and slow processes can be compared.  usage: python tools/bimodal_probe.py [mesh] [pad_bytes]"""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
junk = torch.empty(pad, dtype=torch.uint8, device=dev) if pad else None     # shifts every later allocation
r = bench.Runner(n, 10, dev)
r.run(10)
torch.cuda.synchronize()
names, fwd, bwd = r.profile()
ms = {nm: (fwd[0][i] + bwd[0][i]) / max(fwd[2][i] + bwd[2][i], 1) for i, nm in enumerate(names) if fwd[2][i] + bwd[2][i]}
fm = C.POINTER(C.c_float)()
addr = {"states": r._flat.data_ptr(), "fmesh": r.fmesh.data_ptr(), "xb": r.xb.data_ptr(), "vb": r.vb.data_ptr(),
        "pos_bar": r.pos_bar.data_ptr(), "spec": r.spec.data_ptr()}
free, total = torch.cuda.mem_get_info()
import subprocess, threading


def read_smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        card = json.loads(out)
        card = card[sorted(card)[0]]
        return {k.replace(" clock speed:", "").replace("Temperature (Sensor ", "T(").replace("Current Socket Graphics Package Power (W)", "W"): v
                for k, v in card.items() if any(t in k.lower() for t in ("sclk clock speed", "mclk clock speed", "fclk clock speed", "socclk clock speed", "power", "sensor junction", "sensor memory"))}
    except Exception as e:
        return {"error": repr(e)[:100]}


# clocks WHILE the workload runs: a sampler thread beside ~3 s of steps
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        samples.append(read_smi())


th = threading.Thread(target=sampler)
th.start()
import time
t_end = time.time() + 3.0
while time.time() < t_end:
    r.run(10)
    torch.cuda.synchronize()
stop.set()
th.join()
smi = {"during": samples[:4], "after": read_smi()}
print(json.dumps({"pad": pad, "ms": {k: round(v, 4) for k, v in ms.items()},
                  "addr": {k: hex(v) for k, v in addr.items()},
                  "addr_mod_1G": {k: hex(v % (1 << 30)) for k, v in addr.items()},
                  "free_GiB": round(free / 2 ** 30, 2), "smi_after": smi}), flush=True)
