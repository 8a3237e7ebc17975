"""Does this pool allow a process that has initialised the GPU to start child processes?  (one-off probe)"""
import subprocess, sys, torch
torch.cuda.init()
x = torch.ones(4, device="cuda").sum().item()
print("gpu initialised", x)
try:
    r = subprocess.run([sys.executable, "-c", "print('child ok')"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    print("child rc", r.returncode, r.stdout.strip()[-300:])
except Exception as e:
    print("child launch failed:", repr(e))
