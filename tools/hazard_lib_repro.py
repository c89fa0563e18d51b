"""Where does the concurrency corruption live?  {stand-alone, library} victim x {stand-alone, library} aggressor on two HIP streams.

    python tools/hazard_lib_repro.py N B seconds_per_cell reserve_cus [cells]

victims (stream A, launched back to back, every launch compared bit for bit with a quiet launch of the same inputs, on the device):
  lib   the library's rot_score_kernel through fdipt_igso3_rot_score (B*N residues, random unit quaternions, sigma of t = 0.5)
  own   tools/micro/hazard_repro.hip's float64 series kernel (no library code)
  int   the same kernel with an integer hash chain instead of the float64 series
  valu / bperm / load / lds   victims that isolate one instruction class, checked in registers (hazard_repro.hip: victim2_kernel): VALU only,
        + ds_bpermute butterfly, global loads of a known pattern, LDS write / read round trips; trans / fmath / f64: transcendental unit,
        library float math (division, atan2f, sinf), float64; tid: the work-item id register v0 against the EXEC-derived lane index;
        histogram = 16-lane quarter of the wave
aggressors (stream B):
  fwd   full score-network forwards of a B-sample batch with FdiptForwardArgs.reserve_cus = reserve_cus (the case the soak of round 3 failed in)
  mfma  hazard_repro.hip's persistent MFMA power kernel on 256 - reserve_cus CUs (random operands, with a global read stream)
  dma / ld / dmanw / touch   hazard_repro.hip's memory streamer on 256 - reserve_cus CUs: LDS-DMA (global_load_lds_dwordx4, the library's
        weight-stream idiom) / plain global_load_dwordx4 / LDS-DMA with the wave ending while its last chunk is in flight / unwaited touch loads
  expco   v_exp_f32 chains only (transcendental unit), small blocks: co-resident with the victim
  chainv / chaina / agpr4 / chainaa   co-resident MFMA kernels shaped like the attention's logit phase: ONE dependent accumulator chain in
        VGPRs / in AGPRs, four independent AGPR accumulators, AGPR chain with an AGPR A operand; chainab: B operand from an AGPR; chainaab: both;
        chainaabv: both with the accumulator in VGPRs; accmov: no MFMA, v_accvgpr_read / write only
  ldagpr / ldagprmov / ldvgpr   co-resident kernels whose global loads write AGPRs (then an MFMA with that AGPR as A operand / v_accvgpr_read
        only) or VGPRs (control)
  exec0 / exec1 / exec2 / exec4 / exec8 / execonly / mfmaonly   co-resident kernel: an MFMA followed after 0 / 1 / 2 / 4 / 8 nop cycles by a scalar
        write of EXEC (s_and_saveexec, a masked VALU op, s_or exec); the EXEC sequence alone; the MFMA alone
  mfmaco  the same MFMA kernel with 8 KB of LDS and 256-thread blocks, 2 blocks per CU: victim waves SHARE its SIMDs (co-residency)
  et / ipa / points   one module of the library in a loop through its C-ABI entry (fdipt_edge_transition_fwd: fold rows + EdgeTransition;
        fdipt_ipa_attention_fwd: projection, points, pair bias, attention, o_pair, output projection; fdipt_ipa_project_points: projection + points)
  none  quiet
Prints per cell: victim launches, mismatching launches, mismatching residues and their (index mod 4) histogram.
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from framedipt_amd import _lib, config, sharding  # noqa: E402
from framedipt_amd.diffusion import SE3Diffuser  # noqa: E402
from framedipt_amd.model import ScoreNetwork  # noqa: E402
from framedipt_amd.model.score_network import BatchState  # noqa: E402
from framedipt_amd.sampler import UnconditionalSampler  # noqa: E402

N, B, SEC, RES = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
CELLS = sys.argv[5].split(",") if len(sys.argv) > 5 else ["lib:none", "lib:fwd", "own:fwd", "lib:mfma", "own:mfma"]
lib = _lib.load()
hz = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libhazard.so"))
hz.hz_victim.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
hz.hz_aggressor.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
hz.hz_victim_int.argtypes = hz.hz_victim.argtypes
hz.hz_aggressor_co.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
hz.hz_victim2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
hz.hz_exp_aggressor.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
hz.hz_agpr_load_aggressor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
hz.hz_exec_aggressor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
hz.hz_chain_aggressor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
hz.hz_streamer.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
P = _lib.ptr
conf = config.base_config()
d = SE3Diffuser(conf.diffuser, device="cuda")
net = ScoreNetwork(conf.model, d, precision="fp16").load_synthetic(7).to("cuda")
g = torch.Generator().manual_seed(5)
unit = lambda: torch.nn.functional.normalize(torch.randn(B, N, 4, generator=g), dim=-1).cuda().contiguous()  # noqa: E731
qt, q0 = unit(), unit()
mask = torch.ones(B, N).cuda()
t32, temb, sig = net.step_scalars(np.full(B, 0.5))
sigma = torch.as_tensor(sig, device="cuda")
score, ref = torch.empty(B, N, 3, dtype=torch.float64).cuda(), torch.empty(B, N, 3, dtype=torch.float64).cuda()
# stand-alone victim buffers
hq = torch.tensor([0.9238795, 0.2209424, -0.1913417, 0.2514080, 0.3826834, -0.5334021, 0.6532815, 0.3753303], dtype=torch.float32).cuda()
n_items = B * N
hres, hexp, hbad = torch.empty(n_items * 3 + 8, dtype=torch.float64).cuda(), torch.empty(8, dtype=torch.float64).cuda(), torch.zeros(2048, dtype=torch.int32).cuda()
# aggressor: forwards
ds = UnconditionalSampler(config.to_conf({"min_length": N, "max_length": N, "length_step": 1, "samples_per_length": B}), d, "cuda")
feats, _ = sharding.stack_items([sharding.seeded_item(ds, i, 3, d, 6, 0.01) for i in range(B)])
f32 = lambda x: x.to(device="cuda", dtype=torch.float32).contiguous()  # noqa: E731
ast = BatchState(net, feats["seq_idx"])
ast.reserve_cus = RES
aargs = (f32(feats["rigids_t"]), f32(feats["res_mask"]), f32(feats["fixed_mask"]), f32(feats["sc_ca_t"]) + 1.0, None,
         f32(feats["torsion_angles_sin_cos"][..., 2, :]), torch.as_tensor(t32, device="cuda"), torch.as_tensor(temb, device="cuda"),
         torch.as_tensor(sig, device="cuda"))
# aggressor: MFMA power kernel
ops = torch.randint(0, 1 << 15, (64 * 8 * 64 * 8,), dtype=torch.int16).cuda()  # (positive finite halfs)
gsrc = torch.ones(1 << 28, dtype=torch.int32).cuda()  # 1 GB read stream
aout = torch.zeros(1024).cuda()
# aggressors: single modules through their C-ABI entries (inputs as tools/conc_victim_check.py)
import ctypes as _C  # noqa: E402
mst = BatchState(net, feats["seq_idx"])
m_node = torch.randn(B, N, 256, generator=g).cuda()
m_z = torch.randn(B, N, N, 128, generator=g).cuda().half().contiguous()
m_z2 = torch.empty_like(m_z)
m_rig = torch.cat([unit().cpu(), 10 * torch.randn(B, N, 3, generator=g)], -1).cuda().contiguous()
m_out = torch.empty(B, N, 256).cuda()
m_qp, m_kp, m_vp = torch.empty(B, N, 8, 8, 3).cuda(), torch.empty(B, N, 8, 8, 3).cuda(), torch.empty(B, N, 8, 12, 3).cuda()
dm, pr, dr = _C.byref(net.dims), P(net.params), P(net.derived)


def module(what):
    sp = _lib.stream_ptr()
    if what == "ipa":
        _lib.check(lib.fdipt_ipa_attention_fwd(dm, pr, dr, 1, B, N, P(m_node), P(m_z), P(m_rig), P(mask), P(m_out), P(mst.ws), mst.ws_bytes, sp))
    elif what == "points":
        _lib.check(lib.fdipt_ipa_project_points(dm, pr, dr, 1, B, N, P(m_node), P(m_rig), P(mask), P(m_qp), P(m_kp), P(m_vp), P(mst.ws), mst.ws_bytes, sp))
    else:
        _lib.check(lib.fdipt_edge_transition_fwd(dm, pr, dr, 1, B, N, P(m_node), P(mask), P(m_z), P(m_z2), P(mst.ws), mst.ws_bytes, sp))


CHAINS = {"chainv": 0, "chaina": 1, "agpr4": 2, "chainaa": 3, "chainab": 4, "chainaab": 5, "chainaabv": 6, "accmov": 7}
V2 = {"valu": 0, "bperm": 1, "load": 2, "lds": 3, "trans": 4, "fmath": 5, "f64": 6, "tid": 7}
pat = torch.empty(1 << 22, dtype=torch.int32).cuda()  # 16 MB pattern for the load victim
hz.hz_victim2(0, 0, 0, P(pat), pat.numel(), P(hbad), 1, None)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def victim_lib(out):
    _lib.check(lib.fdipt_igso3_rot_score(B, N, P(qt), P(q0), P(sigma), P(mask), P(out), _lib.stream_ptr()), "rot_score")


def run(cell):
    vic, agg = cell.split(":")
    cnt = torch.zeros(B * N, dtype=torch.int64).cuda()
    launches_bad = torch.zeros((), dtype=torch.int64).cuda()
    with torch.cuda.stream(sa):
        if vic == "lib":
            victim_lib(ref)
        elif vic in V2:
            hbad.zero_()
        else:
            hv = hz.hz_victim_int if vic == "int" else hz.hz_victim
            hbad.zero_()
            hv(n_items, P(hq), 0.6, P(hres), None, P(hbad), _lib.stream_ptr())
            hexp[:3] = hres[:3]
            hbad[:64].zero_()
    torch.cuda.synchronize()
    launches, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < SEC:
        with torch.cuda.stream(sb):
            if agg == "fwd":
                for _ in range(2):
                    ast.forward(*aargs, False)
            elif agg in ("et", "ipa", "points"):
                for _ in range({"et": 3, "ipa": 6, "points": 24}[agg]):
                    module(agg)
            elif agg in ("dma", "ld", "dmanw", "touch"):
                for _ in range(8):
                    hz.hz_streamer({"dma": 0, "ld": 1, "dmanw": 2, "touch": 3}[agg], 256 - RES, 40, P(aout), P(gsrc), _lib.stream_ptr())
            elif agg == "expco":  # transcendental-unit load only, co-resident with the victim
                for _ in range(8):
                    hz.hz_exp_aggressor(1024, 4000, P(aout), _lib.stream_ptr())
            elif agg in ("ldagpr", "ldagprmov", "ldvgpr"):  # global loads into AGPRs (+ MFMA with that AGPR operand / moves only) / into VGPRs (control)
                for _ in range(8):
                    hz.hz_agpr_load_aggressor({"ldagpr": 0, "ldagprmov": 1, "ldvgpr": 2}[agg], 512, 600, P(gsrc), 1 << 22, P(aout), _lib.stream_ptr())
            elif agg.startswith("exec"):  # execG: MFMA, G nop cycles, scalar EXEC write; execonly: no MFMA; mfmaonly: no EXEC write
                # execG: 32x32x16 f16; exec16x16_G: 16x16x32 f16 (4 passes); execf32_G: 32x32x2 f32 (16 passes)
                if agg == "execonly": mode, gap = 1, 0
                elif agg == "mfmaonly": mode, gap = 2, 0
                elif agg.startswith("exechi_"): mode, gap = 5, int(agg[7:])  # mask keeps lanes 48 - 63 enabled
                elif agg.startswith("exec16x16_"): mode, gap = 3, int(agg[10:])
                elif agg.startswith("execf32_"): mode, gap = 4, int(agg[8:])
                else: mode, gap = 0, int(agg[4:])
                for _ in range(8):
                    hz.hz_exec_aggressor(mode, gap, 512, 1500, P(ops), P(aout), _lib.stream_ptr())
            elif agg == "mfmaonly":
                for _ in range(8):
                    hz.hz_exec_aggressor(2, 0, 512, 1500, P(ops), P(aout), _lib.stream_ptr())
            elif agg in CHAINS:  # co-resident MFMA chains: VGPR / AGPR accumulator, 4 AGPR accumulators, AGPR A / B operands, ...
                for _ in range(8):
                    hz.hz_chain_aggressor(CHAINS[agg], 512, 900, P(ops), P(aout), _lib.stream_ptr())
            elif agg == "mfmaco":
                for _ in range(8):
                    hz.hz_aggressor_co(512, 256, 450, P(ops), P(aout), 8192, _lib.stream_ptr())
            elif agg == "mfma":
                for _ in range(8):
                    hz.hz_aggressor(256 - RES, 450, P(ops), P(aout), 1, P(gsrc), gsrc.numel() // 4, _lib.stream_ptr())
        with torch.cuda.stream(sa):
            for _ in range(48):
                if vic in V2:
                    hz.hz_victim2(V2[vic], (n_items + 15) // 16, 8 if vic == "tid" else 512, P(pat), pat.numel(), P(hbad), 0, _lib.stream_ptr())
                elif vic == "lib":
                    victim_lib(score)
                    diff = (score.view(torch.int64) != ref.view(torch.int64)).any(-1).reshape(-1)
                    cnt += diff
                    launches_bad += diff.any()
                else:
                    hv(n_items, P(hq), 0.6, P(hres), P(hexp), P(hbad), _lib.stream_ptr())
            launches += 48
        sa.synchronize()
        sb.synchronize()
    torch.cuda.synchronize()
    if vic == "lib":
        c = cnt.cpu().numpy()
        idx = np.nonzero(c)[0]
        hist = [int(c[(np.arange(c.size) % N) % 4 == k].sum()) for k in range(4)]
        print(f"{cell:10s} N={N} B={B} reserve={RES}: victim launches {launches}, bad launches {int(launches_bad)}, bad residues {int(c.sum())}, "
              f"(residue index mod 4) {hist}, first residues {[(int(i) // N, int(i) % N) for i in idx[:6]]}", flush=True)
    else:
        hb = hbad.cpu().numpy()
        extra = ""
        if vic in ("own", "int") and hb[6]:
            rec = hbad[64:64 + 48].view(torch.int64).cpu().numpy().reshape(8, 3)
            exp_true = hexp[:3].view(torch.int64).cpu().numpy()
            extra = "\n    first mismatches (item, component): value | expectation as loaded | re-loaded at L2 | true expectation:\n" + "\n".join(
                f"      ({int(hb[8 + 4 * k])}, {int(hb[9 + 4 * k])}): {int(rec[k, 0]):#018x} | {int(rec[k, 1]):#018x} | {int(rec[k, 2]):#018x} | {int(exp_true[int(hb[9 + 4 * k])]):#018x}"
                for k in range(min(int(hb[6]), 8)))
        if vic in ("own", "int") and hb[7]:
            rec = hbad[256:256 + 8 * 16 * 6].view(torch.int64).cpu().numpy().reshape(8, 16, 3)
            def row(r):
                return (f"pre {int(r[0]) & 0xFFFFFFFFFFFFFFFF:#018x} omega {int(r[1]) & 0xFFFFFFFF:#010x} trips {(int(r[1]) >> 32) & 0xFFFF} item {int(r[2]) & 0xFFFFFF} "
                        f"tid {(int(r[2]) >> 24) & 0x3FF} hw_id {(int(r[2]) >> 34) & 0xFFFFFFFF:#x}")
            extra += "\n    per-lane records of the first mismatching item (sub 0..15), against item 7 of the quiet launch:\n" + "\n".join(
                f"      sub {k:2d}: BAD {row(rec[0, k])}\n              REF {row(rec[7, k])}" for k in range(16))
        if vic == "tid" and hb[6]:
            extra = "  first (lane | block << 8, v0): " + str([(int(hb[8 + 2 * k]) & 255, int(hb[8 + 2 * k]) >> 8, int(hb[9 + 2 * k])) for k in range(min(int(hb[6]), 8))])
        print(f"{cell:10s} N={N} B={B} reserve={RES}: victim launches {launches}, bad values {int(hb[0])}, (item mod 4 / wave quarter) {[int(x) for x in hb[1:5]]}{extra}", flush=True)


for cell in CELLS:
    run(cell)
