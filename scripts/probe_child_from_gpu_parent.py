"""Can a process that has initialised the GPU start a fresh interpreter as a CHILD (fork + exec in the child)?  bench.py's side
workloads and tests/test_gpu_two_rank_overlap.py rely on it; the pool forbids only exec-replacing the GPU-initialised process."""
import subprocess, sys, torch
torch.zeros(4, device="cuda").sum().item()
r = subprocess.run([sys.executable, "-c", "import torch; print('child sees', torch.cuda.device_count(), 'GPU(s);', float(torch.ones(3, device='cuda').sum()))"],
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print("child rc", r.returncode, "|", r.stdout.strip()[-300:])
