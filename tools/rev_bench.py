"""Micro-benchmark of the fused reverse step (with / without the atom37 frame) and the score launch, N=300, B=8."""
import sys, time
import torch
sys.path.insert(0, ".")
from framedipt_amd import _lib, config, residue_tables
from framedipt_amd.diffusion import SE3Diffuser

d = SE3Diffuser(config.base_config().diffuser)
B, N = 8, 300
g = torch.Generator().manual_seed(5)
q = torch.randn(B, N, 4, generator=g)
t7 = torch.cat([q / q.norm(dim=-1, keepdim=True), 10 * torch.randn(B, N, 3, generator=g)], -1).float().cuda().contiguous()
rs = (0.3 * torch.randn(B, N, 3, generator=g, dtype=torch.float64)).cuda()
ts = (0.1 * torch.randn(B, N, 3, generator=g)).cuda()
dm = torch.ones(B, N).cuda()
zr, zt = torch.randn(B, N, 3, generator=g, dtype=torch.float64).cuda(), torch.randn(B, N, 3, generator=g, dtype=torch.float64).cuda()
psi = torch.nn.functional.normalize(torch.randn(B, N, 2, generator=g), dim=-1).cuda().contiguous()
tb = torch.as_tensor(residue_tables.packed_bytes()).cuda()
a37 = torch.empty(B, N, 37, 3, device="cuda")
out = torch.empty_like(t7)


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("reverse + atoms  %.2f us" % timeit(lambda: d.reverse_device(t7, rs, ts, dm, zr, zt, 0.4, 0.002, True, 0.1, rigids_out=out, atoms=(psi, None, tb, a37))))
print("reverse only     %.2f us" % timeit(lambda: d.reverse_device(t7, rs, ts, dm, zr, zt, 0.4, 0.002, True, 0.1, rigids_out=out)))
print("reverse no mask  %.2f us" % timeit(lambda: d.reverse_device(t7, rs, ts, None, zr, zt, 0.4, 0.002, True, 0.1, rigids_out=out)))
lib = _lib.load()
print("backbone only    %.2f us" % timeit(lambda: _lib.check(lib.fdipt_backbone_atoms(B * N, _lib.ptr(out), None, None, _lib.ptr(psi), None, _lib.ptr(tb), _lib.ptr(a37), None, _lib.stream_ptr()))))
