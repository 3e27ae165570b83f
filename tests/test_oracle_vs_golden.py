"""The CPU oracle (oracle/cvvdp_oracle.py) against vectors produced by the real reference
(oracle/make_goldens.py).  This is what pins the oracle; the GPU tests then compare the HIP
path against the oracle and against the same vectors."""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from oracle import cvvdp_oracle as orc


def _oracle_for(meta, keep=False):
    return orc.Oracle(display_name=meta.get("display"), heatmap=meta["heatmap"], temp_padding=meta["temp_padding"], keep=keep,
                      photometry=meta.get("custom_photometry"), geometry=meta.get("custom_geometry")) \
        if "custom_photometry" not in meta else \
        orc.Oracle(display_name=None, heatmap=meta["heatmap"], temp_padding=meta["temp_padding"], keep=keep,
                   photometry=meta["custom_photometry"], geometry=meta["custom_geometry"])


def _inputs(g):
    t, r = g["test"], g["ref"]
    if t.dtype == np.float16:  # fp16 cases were fed to the reference as torch tensors
        t, r = torch.tensor(t), torch.tensor(r)
    return t, r


@pytest.mark.parametrize("name", golden_cases())
def test_end_to_end(name):
    g = load_golden(name)
    meta = g["meta"]
    o = _oracle_for(meta)
    t, r = _inputs(g)
    jod, stats = o.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    assert abs(o.ppd - float(g["ppd"])) < 1e-9
    np.testing.assert_allclose(stats["rho_band"], g["rho_band"], rtol=1e-12)
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(jod.numpy(), g["jod"], atol=2e-5)
    if "taps" in g:
        np.testing.assert_allclose(np.stack([x.numpy() for x in o.taps]), g["taps"], atol=1e-8)
    if "heatmap" in g:
        hm = stats["heatmap"].numpy().astype(np.float32)
        ref = g["heatmap"].astype(np.float32)
        assert hm.shape == ref.shape
        # fp16 output: allow one fp16 ulp on a tiny fraction of pixels (rounding of equal-to-1e-7 fp32 values)
        bad = np.abs(hm - ref) > 2e-3
        assert bad.mean() < 1e-4, bad.mean()
        assert np.abs(hm - ref).max() < 5e-3


@pytest.mark.parametrize("name", ["img_u8_64x96_fhd_thr", "vid_u8_72x128x12_60_fhd"])
def test_intermediates(name):
    g = load_golden(name)
    meta = g["meta"]
    o = _oracle_for(meta, keep=True)
    t, r = _inputs(g)
    o.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    d = o.dbg
    np.testing.assert_allclose(d["R"].numpy(), g["i_R"], rtol=1e-6, atol=1e-6)
    L = len(d["gpyr"])
    for i in range(L):
        np.testing.assert_allclose(d["gpyr"][i].numpy(), g["i_g%d" % i], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(d["contrast"][i].numpy(), g["i_c%d" % i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(d["logL"][i].numpy(), g["i_l%d" % i], rtol=1e-6, atol=1e-6)
    for i in range(L - 1):  # masking is not applied to the baseband
        np.testing.assert_allclose(d["S"][i].numpy(), g["i_S%d" % i], rtol=1e-5)
        np.testing.assert_allclose(d["D"][i].numpy(), g["i_D%d" % i], rtol=1e-4, atol=1e-6)


def test_setup_vectors(setup_vectors):
    s = setup_vectors
    o = orc.Oracle("standard_4k")
    for fps in (24, 25, 30, 50, 60, 120):
        np.testing.assert_allclose(np.stack([x.numpy() for x in o.temporal_filters(fps)]), s["taps_%d" % fps], atol=1e-8)
    for disp in ("standard_4k", "standard_hdr_pq"):
        d = orc.Display(disp)
        np.testing.assert_array_equal(d.dkl_matrix().numpy(), s["dkl_" + disp])
        np.testing.assert_allclose(np.array(d.black_level()), s["black_" + disp], rtol=1e-15)
    for i, rho in enumerate(s["csf_rhos"]):
        for j, (oo, cc) in enumerate(((0, 0), (0, 1), (0, 2), (1, 0))):
            np.testing.assert_allclose(o.csf.row(float(rho), oo, cc).numpy(), s["csf_rows"][i, j], rtol=1e-6, atol=1e-7)
    q = torch.tensor(s["csf_query"])
    for j, (oo, cc) in enumerate(((0, 0), (0, 1), (0, 2), (1, 0))):
        np.testing.assert_allclose(o.csf.sensitivity(o.csf.row(3.0424, oo, cc), q).numpy(), s["csf_S"][j], rtol=1e-6)
    for row, disp in zip(s["band_sizes"], s["band_displays"]):
        W, H, ppd, nb = int(row[0]), int(row[1]), row[2], int(row[3])
        d = orc.Display(str(disp))
        assert abs(d.ppd - ppd) < 1e-9
        h, fr = orc.band_frequencies(W, H, d.ppd)
        assert h + 1 == nb
        np.testing.assert_allclose(fr, row[4:4 + nb], rtol=1e-12)


def test_photometry_ramps(setup_vectors):
    s = setup_vectors
    ramp = torch.tensor(s["fwd_ramp"])
    for disp in ("standard_4k", "standard_hdr_pq", "standard_hdr_hlg", "standard_hdr_linear", "standard_phone"):
        d = orc.Display(disp)
        inp = ramp * (1000.0 if disp == "standard_hdr_linear" else 1.0)
        np.testing.assert_allclose(d.to_dkl(inp).numpy(), s["fwd_" + disp], rtol=1e-6, atol=1e-6)
    d = orc.Display(photometry=dict(Y_peak=120, contrast=800, source_colorspace="Adobe RGB (1998)", E_ambient=80), geometry=dict(resolution=(1920, 1200), ppd=60))
    np.testing.assert_allclose(d.to_dkl(ramp).numpy(), s["fwd_gamma22"], rtol=1e-6, atol=1e-6)
    d = orc.Display(photometry=dict(Y_peak=300, contrast=2000, source_colorspace="sRGB", E_ambient=10, exposure=0.7), geometry=dict(resolution=(1920, 1200), ppd=60))
    np.testing.assert_allclose(d.to_dkl(ramp).numpy(), s["fwd_srgb_exp07"], rtol=1e-6, atol=1e-6)


def test_oracle_fullsize_prefix_against_reference():
    """The oracle on 1080p frames of bench.py's synthetic clip against the real reference (the 4K case runs in the GPU suite)."""
    import pytest
    from conftest import fullsize_inputs, load_golden
    from oracle import cvvdp_oracle as orc
    g = load_golden("fullsize_fhd_4f")
    inp = fullsize_inputs(g)
    if inp is None:
        pytest.skip("this torch build's CPU generator does not reproduce the fixture's synthetic frames")
    jod, stats = orc.Oracle(str(g["display"])).predict(inp[0], inp[1], dim_order="BCFHW", frames_per_second=float(g["fps"]))
    assert abs(float(jod) - float(g["jod"])) <= 1e-4
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-5, atol=2e-7)


def test_oracle_reproduces_the_documented_known_answer():
    """examples/ex_simple_image.py:14-17 documents 'Blur - Quality: 8.514 JOD' for wavy_facade.png (16-bit) on standard_4k."""
    from conftest import kat_wavy_facade
    g, test, ref = kat_wavy_facade()
    jod, stats = orc.Oracle("standard_4k").predict(test, ref, dim_order="HWC")
    assert round(float(jod), 3) == round(float(g["documented_jod"]), 3) == 8.514
    assert abs(float(jod) - float(g["jod"])) <= 2e-5           # the real reference on the same samples: 8.51376
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-5, atol=2e-7)


def test_oracle_reproduces_the_documented_hdr_known_answer():
    """examples/ex_hdr_images.py:13-17 documents 'Blur - Quality: 8.696 JOD' for nancy_church.hdr on a linear-EOTF 4000 cd/m^2 display."""
    from conftest import kat_nancy_church
    g, test, ref, photo = kat_nancy_church()
    jod, stats = orc.Oracle(display_name="standard_hdr_linear", photometry=photo).predict(test, ref, dim_order="HWC")
    assert round(float(jod), 3) == round(float(g["documented_jod"]), 3) == 8.696
    assert abs(float(jod) - float(g["jod"])) <= 2e-5
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-5, atol=2e-7)


def test_oracle_features_against_reference():
    """The oracle's restatement of the ML heads' feature pooling (cvvdp_ml_metric.py:77-107, :351-358) against the real reference's
    extract_features (tests/golden/features.npz, oracle/make_goldens_features.py): the GPU suite uses it on shapes the fixtures
    do not cover (ragged widths, several strips and segments)."""
    from conftest import load_golden
    from oracle import cvvdp_oracle as orc
    gf = load_golden("features")
    k = 0
    while f"case{k}" in gf:
        g = load_golden(str(gf[f"case{k}"]))
        meta = g["meta"]
        o = orc.Oracle(display_name=meta["display"], temp_padding=meta["temp_padding"], features=True)
        _, stats = o.predict(g["test"], g["ref"], dim_order=meta["dim_order"], frames_per_second=meta["fps"])
        feats = stats["features"]
        assert len(feats) == int(gf[f"case{k}_bands"])
        for bb, f in enumerate(feats):
            want = gf[f"case{k}_band{bb}"]
            assert f.shape == want.shape, (bb, f.shape, want.shape)
            for q in (0, 2, 4):
                np.testing.assert_allclose(f[..., q], want[..., q], rtol=2e-5, atol=2e-7, err_msg=f"band {bb} mean {q}")
                scale = np.abs(want[..., q]) ** 2 + np.abs(want[..., q + 1])
                assert np.all(np.abs(f[..., q + 1] - want[..., q + 1]) <= 1e-4 * scale + 1e-8), f"band {bb} var {q + 1}"
        k += 1
    assert k == 2


def test_long_8k_fixtures_agree_where_they_overlap():
    """The real reference's scores of configs[4]'s clip at 17 (heat map), 64 (heat map), 80 and 256 frames (oracle/make_goldens_8k17_pqrange.py,
    _8k64_heat.py, _8k80.py [256]): four separate runs of its CPU path on prefixes of ONE clip.  The temporal filter is causal, so a prefix's
    per-frame scores must be the longer run's, bit for bit -- the claim every test that holds a long GPU run against a short fixture rests on."""
    import os
    import numpy as np
    from conftest import GOLDEN, load_golden
    import glob
    have = [n for n in ("deep_8k_pqrange_heat_17f", "deep_8k_pqrange_heat_64f") if os.path.isfile(os.path.join(GOLDEN, n + ".npz"))]
    import re
    have += sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "deep_8k_pq_*f.npz"))
                   if re.search(r"deep_8k_pq_\d+f\.npz$", p))                                                   # 80 frames, and what the resumable generator has finished
    assert "deep_8k_pq_80f" in have and "deep_8k_pqrange_heat_17f" in have
    gs = [load_golden(n) for n in have]
    for a in gs:
        for b in gs:
            fa, fb = int(a["frames"]), int(b["frames"])
            if fa < fb:
                np.testing.assert_array_equal(a["Q_per_ch"], b["Q_per_ch"][:, :, :fa])
                np.testing.assert_array_equal(a["rho_band"], b["rho_band"])
                if "heatmap_frame_means" in a and "heatmap_frame_means" in b:
                    # (the generators take a frame's mean over a strided slice of the run's own [1,3,F,H,W] tensor: torch's reduction order
                    # depends on F, so one fp32 ulp of the mean is the generators', not the reference's -- the down-sampled frames below are bits)
                    np.testing.assert_allclose(a["heatmap_frame_means"], b["heatmap_frame_means"][:fa], rtol=3e-7, atol=0)
                if "heatmap_ds" in a and "heatmap_ds" in b:
                    # the frames both runs kept, on the samples both kept (every 16th pixel in the 17-frame fixture, every 24th in the 64-frame one)
                    sa, sb = (int(x["heatmap_ds_step"]) if "heatmap_ds_step" in x else 16 for x in (a, b))
                    lcm = int(np.lcm(sa, sb))
                    fa_k, fb_k = [int(k) for k in a["heatmap_frames"]], [int(k) for k in b["heatmap_frames"]]
                    for k in set(fa_k) & set(fb_k):
                        np.testing.assert_array_equal(a["heatmap_ds"][:, fa_k.index(k), ::lcm // sa, ::lcm // sa], b["heatmap_ds"][:, fb_k.index(k), ::lcm // sb, ::lcm // sb])
