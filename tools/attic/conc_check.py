import torch, time
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1): torch.cuda._sleep(200_000_000)
    with torch.cuda.stream(s2 if two else s1): torch.cuda._sleep(200_000_000)
    torch.cuda.synchronize(); return time.perf_counter() - t0
run(True)
print("same stream", run(False), "two streams", run(True))
import os; print({k: v for k, v in os.environ.items() if "HIP" in k or "HSA" in k or "GPU_" in k or "ROC" in k})
