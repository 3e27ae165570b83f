"""Where does the Q_per_ch deviation of the thin class come from?  (round 6; run on the GPU box)
    python tools/hybrid_level_experiment.py fuzz_seed21_case35 [more fixture names | seed:case]

For a clip of the randomised sweep: the GPU's own Gaussian pyramid (debug dump, unfused route) is handed to the ORACLE level by level --
"hybrid k" = the GPU's levels 0..k, every level below k and ALL band arithmetic (expand, contrast, CSF, masking, pooling) by the oracle
(= the reference's torch operators).  hybrid 0 isolates what the GPU's level-0 planes (display model + temporal filter) contribute, the step
from hybrid k-1 to hybrid k what the GPU's reduce of level k-1 adds, and "gpu" - "hybrid L-1" what the GPU's band kernels add.  Errors are the
worst Q_per_ch entry against the fixture of the real reference (or the oracle for seed:case arguments), in units of the test tolerance."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import colorvideovdp_amd as cv
from colorvideovdp_amd import _capi
from oracle import cvvdp_oracle as orc
from tools import fuzz_cases


def err(q, qr):
    q, qr = np.float64(q), np.float64(qr)
    e = np.abs(q - qr) / (2e-4 * np.abs(qr) + 2e-6)
    return e.max(), e.max(axis=(0, 1, 2))


for arg in sys.argv[1:]:
    if ":" in arg:
        seed, k = (int(x) for x in arg.split(":"))
        g = None
    else:
        from conftest import load_golden
        g = load_golden(arg)
        seed, k = int(g["seed"]), int(g["case"])
    c = [c for c in fuzz_cases.cases(seed, k + 1, only={k})][0]
    test, ref = fuzz_cases.as_input(c["test"]), fuzz_cases.as_input(c["ref"])
    o = orc.Oracle(display_name=c["display"], temp_padding=c["padding"], heatmap=None)
    oj, os_ = o.predict(test, ref, dim_order="BCFHW", frames_per_second=c["fps"])
    qref = g["Q_per_ch"] if g is not None else os_["Q_per_ch"]
    print(f"== {arg}: {c['W']}x{c['H']}x{c['F']} B{c['B']} {c['dtype']} C{test.shape[1]} fps {c['fps']} {c['display']} {c['padding']}; oracle vs reference: {err(os_['Q_per_ch'], qref)[0]:.3f}")
    if c["B"] != 1 or c["F"] < 2:
        print("   (batch / image case: skipped)")
        continue
    m = cv.cvvdp(display_name=c["display"], temp_padding=c["padding"], block_frames=c["F"])
    m.debug_dump, m.fuse_mode = True, 2
    j, s = m.predict(test, ref, dim_order="BCFHW", frames_per_second=c["fps"])
    L, F = len(s["rho_band"]), c["F"]
    sizes = [(c["H"], c["W"])]
    for _ in range(1, L):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    gp = [m.debug_buffer(_capi.BUF_GPYR, l).cpu().reshape(8, -1, *sizes[l])[:, :F].clone() for l in range(L)]
    print("   gpu                     : %.3f  per band %s" % (err(s["Q_per_ch"], qref)[0], np.round(err(s["Q_per_ch"], qref)[1], 2)))
    m1 = cv.cvvdp(display_name=c["display"], temp_padding=c["padding"], block_frames=c["F"])
    m1.fuse_mode = 1
    _, s1 = m1.predict(test, ref, dim_order="BCFHW", frames_per_second=c["fps"])
    print("   gpu, fused route        : %.3f  per band %s" % (err(s1["Q_per_ch"], qref)[0], np.round(err(s1["Q_per_ch"], qref)[1], 2)))
    rho = s["rho_band"]
    orig = orc.gaussian_pyramid
    for kk in range(L):
        Q = np.zeros_like(qref)
        for f in range(F):
            def gpyr(R, levels, kk=kk, f=f):
                gl = [gp[l][:, f][None, :, None] for l in range(kk + 1)]
                while len(gl) < levels:
                    gl.append(orc.pyr_reduce(gl[-1]))
                return gl
            orc.gaussian_pyramid = gpyr
            Qb, _ = o.process_block(gp[0][:, f][None, :, None], L, rho, False)
            Q[:, :, f] = Qb.numpy()[:, :, 0]
        orc.gaussian_pyramid = orig
        e = err(Q, qref)
        print("   hybrid %d (gpu levels 0..%d, %4dx%-4d): %.3f  per band %s" % (kk, kk, sizes[kk][1], sizes[kk][0], e[0], np.round(e[1], 2)))
