"""HIP path (through the C ABI) against the reference vectors and the CPU oracle.  Needs an MI355X.

Tolerances (fp32 path, different summation order and libm than torch-CPU):
  JOD        |d| <= 1e-3   (north-star bound; observed ~1e-5)
  Q_per_ch   rtol 2e-4, atol 2e-6
  heat map   fp16 output: <= 1e-3 of the pixels may differ by more than 2e-3, none by more than 2e-2
"""
import os

import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden

pytestmark = pytest.mark.gpu

JOD_TOL = 1e-3


def _metric(meta, **kw):
    import colorvideovdp_amd as cv
    if "custom_photometry" in meta:
        ph = cv.vvdp_display_photo_eotf(**meta["custom_photometry"])
        ge = cv.vvdp_display_geometry(**meta["custom_geometry"])
        return cv.cvvdp(display_photometry=ph, display_geometry=ge, heatmap=meta["heatmap"], temp_padding=meta["temp_padding"], **kw)
    return cv.cvvdp(display_name=meta["display"], heatmap=meta["heatmap"], temp_padding=meta["temp_padding"], **kw)


def _inputs(g):
    t, r = g["test"], g["ref"]
    if t.dtype == np.float16:
        return torch.tensor(t), torch.tensor(r)
    return t, r


def _release(m):
    """Hand a metric's workspace (up to ~100 GB for 8K clips) back to the driver NOW: dropping the last name is not enough while a
    reference cycle keeps the object until the next collection."""
    import gc
    m._ws = None
    m._clip_cache = None
    gc.collect()
    torch.cuda.empty_cache()


def _check_heatmap(got, want, name=None):
    """Generic bound: <= 1e-3 of the pixels off by more than 2e-3, none by more than 2e-2 (~20 fp16 ulp near 1).  Fixtures with a
    recorded margin (conftest.observed_bound) are held to 1.5 x what this build was observed to do on them -- a one-bin shift of the
    tone-map histogram (visualize_diff_map.py:26-35) moves thousands of pixels and cannot hide in the generic bound then."""
    from conftest import observed_bound, record_observed
    got = got.numpy().astype(np.float32) if torch.is_tensor(got) else np.asarray(got, dtype=np.float32)
    want = np.asarray(want).astype(np.float32)
    assert got.shape == want.shape
    d = np.abs(got - want)
    frac, mx, mean = float((d > 2e-3).mean()), float(d.max()), float(d.mean())
    if name is not None:
        record_observed("heatmap", name, {"frac_gt_2e-3": frac, "max": mx, "mean": mean})
    assert frac < 1e-3, frac
    assert mx < 2e-2, mx
    b = observed_bound("heatmap", name) if name is not None else None
    if b is not None:
        assert frac <= b["frac_gt_2e-3"] and mx <= b["max"] and mean <= b["mean"], (name, frac, mx, mean, b)


@pytest.mark.parametrize("name", golden_cases())
def test_golden(name):
    g = load_golden(name)
    meta = g["meta"]
    m = _metric(meta)
    t, r = _inputs(g)
    jod, stats = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    assert jod.device.type == "cuda"
    assert stats["Q_per_ch"].dtype == np.float32 and stats["Q_per_ch"].shape == g["Q_per_ch"].shape
    np.testing.assert_allclose(stats["rho_band"], g["rho_band"], rtol=1e-12)
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(jod.cpu().numpy(), g["jod"], atol=JOD_TOL)
    assert m.get_info_string() == str(g["info"])
    if "heatmap" in g:
        assert stats["heatmap"].dtype == torch.float16 and stats["heatmap"].device.type == "cpu"
        _check_heatmap(stats["heatmap"], g["heatmap"], name)


@pytest.mark.parametrize("name", ["vid_u8_72x128x12_60_fhd", "vid_u16_67x121x20_30_4k_sym", "vid_u8_135x240x18_60_fhd_raw"])
@pytest.mark.parametrize("block", [1, 5, 7])
def test_block_size_invariance(name, block):
    """ChangeLog v0.5.3: results must not depend on how many frames are processed at once."""
    g = load_golden(name)
    meta = g["meta"]
    t, r = _inputs(g)
    jod0, s0 = _metric(meta).predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    jod1, s1 = _metric(meta, block_frames=block).predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    np.testing.assert_array_equal(s0["Q_per_ch"], s1["Q_per_ch"])  # same kernels, same per-frame arithmetic
    assert float(jod0) == float(jod1)
    if "heatmap" in s0:
        assert torch.equal(s0["heatmap"], s1["heatmap"])


def test_intermediates_against_oracle():
    """Ring (DKL), temporal channels, Gaussian pyramid and per-pixel D of every band vs the CPU oracle."""
    from colorvideovdp_amd import _capi
    from oracle import cvvdp_oracle as orc
    g = load_golden("vid_u8_72x128x12_60_fhd")
    meta = g["meta"]
    m = _metric(meta)
    m.debug_dump = True
    t, r = _inputs(g)
    m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    o = orc.Oracle(display_name=meta["display"], keep=True)
    o.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    d = o.dbg  # intermediates of the last frame
    nfr, L = 12, len(d["gpyr"])
    for l in range(L):
        H, W = d["gpyr"][l].shape[-2:]
        buf = m.debug_buffer(_capi.BUF_GPYR, l).cpu().numpy().reshape(8, -1, H, W)
        np.testing.assert_allclose(buf[:, nfr - 1], d["gpyr"][l][0, :, 0].numpy(), rtol=2e-5, atol=2e-5)
        dd = m.debug_buffer(_capi.BUF_DDUMP, l).cpu().numpy().reshape(4, -1, H, W)
        np.testing.assert_allclose(dd[:, nfr - 1], d["D"][l][0, :, 0].numpy(), rtol=5e-4, atol=2e-5)


@pytest.mark.parametrize("case", ["vid_u8_72x128x12_60_fhd", "vid_u8_135x240x18_60_fhd_raw"])
def test_small_levels_reduce_like_the_reference_bit_for_bit(case):
    """Levels of at most 128 x 128 samples are reduced with the reference's operation order (pyramid.hip k_reduce: vertical pass first,
    every pass an in-order FMA chain as torch's CPU conv2d runs it, the edge terms as separately rounded tensor operations): the
    reference's own reduce (the oracle's pyr_reduce = lpyr_dec.py:186-211 on torch) applied to the GPU's level l gives the GPU's level
    l+1 BIT FOR BIT.  The larger levels go through the marching kernels (horizontal pass first): same operator, other roundings.
    Why it matters: profiles/r06_order_experiment.txt (the two coarsest Laplacian bands of luminance-only clips)."""
    from colorvideovdp_amd import _capi
    from oracle import cvvdp_oracle as orc
    g = load_golden(case)
    meta = g["meta"]
    m = _metric(dict(meta, heatmap="none"))
    m.debug_dump = True
    m.fuse_mode = 2
    t, r = _inputs(g)
    _, stats = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    L = len(stats["rho_band"])
    H, W = stats["height"], stats["width"]
    sizes = [(H, W)]
    for _ in range(1, L):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    exact = 0
    for l in range(L - 1):
        (h0, w0), (h1, w1) = sizes[l], sizes[l + 1]
        nfr = int(stats["N_frames"])
        a = m.debug_buffer(_capi.BUF_GPYR, l).cpu().reshape(8, -1, h0, w0)[:, :nfr]      # (items beyond the clip's frames were never written)
        b = m.debug_buffer(_capi.BUF_GPYR, l + 1).cpu().reshape(8, -1, h1, w1)[:, :nfr]
        want = orc.pyr_reduce(a.clone())
        if h0 * w0 <= 16384 or w0 < 16 or h0 < 4:
            assert torch.equal(want, b), (l, (h0, w0), float((want - b).abs().max()))
            exact += 1
        else:
            np.testing.assert_allclose(b.numpy(), want.numpy(), rtol=1e-6, atol=2e-6 * max(1.0, float(want.abs().max())))      # (a few ulps of the level's largest samples)
    assert exact >= 3


def test_frame_shards_are_exact():
    """Frame-range sharding with a real halo reproduces the unsharded per-frame features bit for bit."""
    from colorvideovdp_amd.sharding import plan_frame_shard
    from colorvideovdp_amd.video_source import video_source_array
    g = load_golden("vid_u8_135x240x18_60_fhd_raw")
    meta = g["meta"]
    t, r = _inputs(g)
    m = _metric(dict(meta, heatmap="none"))
    _, s0 = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    vs = video_source_array(t, r, meta["fps"], dim_order=meta["dim_order"])
    for world in (2, 3):
        parts = []
        for rank in range(world):
            first, count = plan_frame_shard(18, rank, world)
            q, _, _ = m._score_range(vs, first, count)
            parts.append(q.cpu().numpy())
        np.testing.assert_array_equal(np.concatenate(parts, axis=2), s0["Q_per_ch"])


def test_device_resident_and_strided_inputs():
    """Inputs already on the GPU, in a non-default dim order, are consumed in place."""
    g = load_golden("vid_u8_72x128x12_60_fhd")
    meta = g["meta"]
    t, r = _inputs(g)
    m = _metric(meta)
    jod0, s0 = m.predict(t, r, dim_order="FHWC", frames_per_second=60)
    tg = torch.tensor(t).cuda().permute(3, 0, 1, 2).contiguous()  # CFHW on device
    rg = torch.tensor(r).cuda().permute(3, 0, 1, 2).contiguous()
    jod1, s1 = m.predict(tg, rg, dim_order="CFHW", frames_per_second=60)
    np.testing.assert_array_equal(s0["Q_per_ch"], s1["Q_per_ch"])
    tf = (torch.tensor(t).float() / 255).cuda()
    rf = (torch.tensor(r).float() / 255).cuda()
    jod2, s2 = m.predict(tf, rf, dim_order="FHWC", frames_per_second=60)
    # u8 samples are scaled by the rounded reciprocal of 255 on the GPU (<= 1 ulp from x/255)
    np.testing.assert_allclose(s2["Q_per_ch"], s0["Q_per_ch"], rtol=1e-4, atol=1e-6)


def test_custom_video_source_slow_path():
    """A user video_source that returns DKL frames (the reference protocol) still works."""
    import colorvideovdp_amd as cv
    from oracle import cvvdp_oracle as orc
    g = load_golden("vid_u8_36x64x9_60_sym_short")
    meta = g["meta"]
    t, r = _inputs(g)
    disp = orc.Display("standard_fhd")
    tt, rr = orc.to_bcfhw(t, "FHWC"), orc.to_bcfhw(r, "FHWC")

    class Src(cv.video_source):
        def get_video_size(self):
            return (36, 64, 9)

        def get_frames_per_second(self):
            return 60

        def get_test_frame(self, frame, device, colorspace):
            assert colorspace == "DKLd65"
            return disp.to_dkl(orc.fetch_frame(tt, frame)).to(device)

        def get_reference_frame(self, frame, device, colorspace):
            return disp.to_dkl(orc.fetch_frame(rr, frame)).to(device)

    m = _metric(meta)
    jod, stats = m.predict_video_source(Src())
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(jod.cpu().numpy(), g["jod"], atol=JOD_TOL)


def test_errors_match_reference_behaviour():
    import colorvideovdp_amd as cv
    m = cv.cvvdp(display_name="standard_fhd", heatmap="threshold")
    x = np.zeros((2, 3, 1, 16, 16), dtype=np.float32)
    with pytest.raises(cv.vq_exception):
        m.predict(x, x, dim_order="BCFHW")  # heat map with a batch
    m = cv.cvvdp(display_name="standard_fhd")
    with pytest.raises(RuntimeError):
        m.predict(np.zeros((3, 4, 16, 16), np.float32), np.zeros((3, 4, 16, 16), np.float32), dim_order="CFHW")  # video without fps
    with pytest.raises(RuntimeError):
        m.predict(np.zeros((16, 16, 2), np.float32), np.zeros((16, 16, 2), np.float32), dim_order="HWC")  # 2 channels
    with pytest.raises(RuntimeError):
        m.predict(np.zeros((16, 16, 3), np.float64), np.zeros((16, 16, 3), np.float64), dim_order="HWC")  # dtype
    with pytest.raises(RuntimeError):
        cv.cvvdp(display_name="no_such_display")


def test_full_size_properties():
    """BASELINE-size frames (1080p) through size-independent properties: identical clips score exactly
    10 JOD with Q_per_ch == 0; a static clip has an exactly-zero transient channel after the filter has
    settled on replicate padding; per-frame features do not depend on the block size."""
    import colorvideovdp_amd as cv
    gen = torch.Generator(device="cuda").manual_seed(7)
    ref = (torch.rand((1, 3, 1, 1080, 1920), generator=gen, device="cuda") * 255).to(torch.uint8).expand(1, 3, 20, 1080, 1920)
    m = cv.cvvdp(display_name="standard_fhd")
    jod, stats = m.predict(ref, ref, dim_order="BCFHW", frames_per_second=60)
    assert float(jod) == 10.0
    assert np.all(stats["Q_per_ch"] == 0)
    noise = (torch.rand((1, 3, 1, 1080, 1920), generator=gen, device="cuda") * 12).to(torch.uint8)
    test = torch.clamp(ref[:, :, :1].to(torch.int16) + noise.to(torch.int16) - 6, 0, 255).to(torch.uint8).expand(1, 3, 20, 1080, 1920)
    jod, stats = m.predict(test, ref, dim_order="BCFHW", frames_per_second=60)
    q = stats["Q_per_ch"]
    assert 5.0 < float(jod) < 10.0
    # static content: every frame sees the same window -> identical features in all frames
    np.testing.assert_allclose(q[:, :, 1:], q[:, :, :1].repeat(19, axis=2), rtol=1e-5, atol=1e-7)
    m2 = cv.cvvdp(display_name="standard_fhd", block_frames=3)
    _, s2 = m2.predict(test, ref, dim_order="BCFHW", frames_per_second=60)
    np.testing.assert_array_equal(s2["Q_per_ch"], q)


def test_heatmap_host_buffer_is_reused_only_when_free():
    """The page-locked heat-map buffer of the previous call is recycled, but never while a caller still holds it."""
    import colorvideovdp_amd as cv
    g = load_golden("vid_u8_135x240x18_60_fhd_raw")
    met = _metric(g["meta"])
    t, r = _inputs(g)
    kw = dict(dim_order=g["meta"]["dim_order"], frames_per_second=g["meta"]["fps"])
    _, s1 = met.predict(t, r, **kw)
    h1 = s1["heatmap"]
    keep = h1.clone()
    _, s2 = met.predict(r, r, **kw)                       # h1 is alive: must not be overwritten
    assert s2["heatmap"].data_ptr() != h1.data_ptr()
    assert torch.equal(h1, keep)
    p2 = s2["heatmap"].data_ptr()
    view = s2["heatmap"][0, 0, 3]                          # a view alone keeps the buffer busy
    del s2
    _, s3 = met.predict(t, r, **kw)
    assert s3["heatmap"].data_ptr() != p2 and torch.equal(s3["heatmap"], keep)
    p3 = s3["heatmap"].data_ptr()
    del s3, view
    _, s4 = met.predict(t, r, **kw)                       # nothing refers to the last buffer any more: recycled
    assert s4["heatmap"].data_ptr() == p3 and torch.equal(s4["heatmap"], keep)


@pytest.mark.parametrize("seed", range(8))
def test_random_shapes_against_oracle(seed):
    """Seeded sweep over sizes that hit every kernel variant (widths that are / are not multiples of 4, 8 and 16, odd
    heights, one- and two-level reduce passes, rotating / chunked / generic FIR), padding modes and displays."""
    import colorvideovdp_amd as cv
    from oracle import cvvdp_oracle as orc
    rng = np.random.default_rng(1000 + seed)
    H, W = [(33, 47), (48, 64), (100, 180), (127, 256), (72, 336), (135, 240), (64, 100), (90, 160)][seed]
    F, fps = [(1, 0), (2, 24), (5, 30), (20, 60), (7, 120), (19, 50), (4, 100), (33, 60)][seed]
    disp = ["standard_fhd", "standard_4k", "standard_hdr_pq", "standard_fhd", "standard_4k", "standard_hdr_hlg", "standard_fhd", "standard_4k"][seed]
    pad = "symmetric" if seed % 3 == 1 else "replicate"
    y, x = np.mgrid[0:H, 0:W]
    ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (2.3 * x / W + f / 17.0) + c) * np.cos(2 * np.pi * 1.7 * y / H) for c in range(3)])
                    for f in range(F)], axis=1)[None]                                   # [1,3,F,H,W]
    test = np.clip(ref + 0.04 * rng.standard_normal(ref.shape) * (x > W // 3), 0, 1)
    if seed % 2:
        test, ref = np.round(test * 255).astype(np.uint8), np.round(np.clip(ref, 0, 1) * 255).astype(np.uint8)
    else:
        test, ref = test.astype(np.float32), np.clip(ref, 0, 1).astype(np.float32)
    try:
        o = orc.Oracle(display_name=disp, temp_padding=pad)
    except Exception:
        pytest.skip(f"display {disp} not in the shipped display models")
    ojod, ostats = o.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    jod, stats = cv.cvvdp(display_name=disp, temp_padding=pad).predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    assert abs(float(jod) - float(ojod)) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], ostats["Q_per_ch"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("name", __import__("conftest").fullsize_cases())
def test_fullsize_prefix_against_reference(name):
    """BASELINE-size frames (4K, 1080p) against outputs of the real reference on a prefix of bench.py's clip."""
    import colorvideovdp_amd as cv
    from conftest import fullsize_inputs
    g = load_golden(name)
    inp = fullsize_inputs(g)
    if inp is None:
        pytest.fail("this torch build's CPU generator does not reproduce the fixture's synthetic frames (checksum mismatch): the BASELINE-size parity check cannot run -- regenerate tests/golden with oracle/make_goldens_fullsize.py / make_goldens_bench.py")
    jod, stats = cv.cvvdp(display_name=str(g["display"])).predict(inp[0], inp[1], dim_order="BCFHW", frames_per_second=float(g["fps"]))
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(stats["rho_band"], g["rho_band"], rtol=1e-12)


def test_block_size_invariance_device_resident():
    """Device-resident clips take the raw-halo route between blocks (no DKL tail): still bit-exact for any block size,
    for both padding modes (blocks shorter than the filter re-read padded frames)."""
    import colorvideovdp_amd as cv
    for case in ("vid_u8_135x240x18_60_fhd_raw", "vid_u16_67x121x20_30_4k_sym"):
        g = load_golden(case)
        meta = dict(g["meta"]); meta["heatmap"] = None
        t, r = _inputs(g)
        t = torch.as_tensor(t.view(np.int16) if isinstance(t, np.ndarray) and t.dtype == np.uint16 else t).cuda()
        r = torch.as_tensor(r.view(np.int16) if isinstance(r, np.ndarray) and r.dtype == np.uint16 else r).cuda()
        qs = []
        for nb in (None, 7, 3, 1):
            _, st = _metric(meta, block_frames=nb).predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
            qs.append(st["Q_per_ch"])
        np.testing.assert_allclose(qs[0], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
        for q in qs[1:]:
            np.testing.assert_array_equal(qs[0], q)


@pytest.mark.parametrize("fuse_mode", [0, 1])
def test_blocks_longer_than_the_nominal_64_frames(fuse_mode):
    """Raw clips resident in HBM are scored in blocks of up to 240 frames (cvvdp_metric.py::_pick_block_frames); the core sizes its
    segments and picks its kernels from a nominal 64-frame block, so a 170-frame clip in one block, in blocks of 100, 64 and 7 gives
    the same bits -- on the reduce + k_band4 route and on the fused band kernels."""
    import colorvideovdp_amd as cv
    t, r = _fuse_clip(256, 144, 170, 11)
    t, r = torch.as_tensor(t).cuda(), torch.as_tensor(r).cuda()
    qs, blocks = [], []
    for nb in (None, 100, 64, 7):
        m = cv.cvvdp(display_name="standard_fhd", block_frames=nb)
        m.fuse_mode = fuse_mode
        _, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=30)
        qs.append(st["Q_per_ch"]); blocks.append(m.last_block_frames)
        assert m.fused_levels == (3 if fuse_mode else 0)
    assert blocks == [170, 100, 64, 7]
    for q in qs[1:]:
        np.testing.assert_array_equal(qs[0], q)


def test_workspace_that_cannot_be_had_falls_back_to_shorter_blocks(monkeypatch):
    """The block length is sized from the memory seen free a moment before the allocation; when the workspace of a long block
    cannot be had any more (several ranks on one GPU), the class retries with 64-frame blocks and below instead of failing --
    with the same bits.  A pinned block_frames is the caller's decision and still raises."""
    import colorvideovdp_amd as cv
    t, r = _fuse_clip(256, 144, 170, 11)
    t, r = torch.as_tensor(t).cuda(), torch.as_tensor(r).cuda()
    m0 = cv.cvvdp(display_name="standard_fhd")
    _, s0 = m0.predict(t, r, dim_order="BCFHW", frames_per_second=30)
    assert m0.last_block_frames == 170
    need170 = m0._ws.numel()
    m = cv.cvvdp(display_name="standard_fhd")
    sizes = []

    def stingy(nbytes, _orig=m._alloc_workspace):
        sizes.append(nbytes)
        if nbytes > need170 // 4:                                  # room for a quarter of the 170-frame workspace only
            raise torch.cuda.OutOfMemoryError("simulated: workspace of %d bytes" % nbytes)
        return _orig(nbytes)

    monkeypatch.setattr(m, "_alloc_workspace", stingy)
    _, s1 = m.predict(t, r, dim_order="BCFHW", frames_per_second=30)
    assert m.last_block_frames == 32 and len(sizes) == 3 and sizes[0] == need170      # 170 -> 64 -> 32
    np.testing.assert_array_equal(s0["Q_per_ch"], s1["Q_per_ch"])
    mp = cv.cvvdp(display_name="standard_fhd", block_frames=100)
    monkeypatch.setattr(mp, "_alloc_workspace", stingy)
    with pytest.raises(torch.cuda.OutOfMemoryError):
        mp.predict(t, r, dim_order="BCFHW", frames_per_second=30)


def test_documented_known_answer():
    """The reference's own KAT (examples/ex_simple_image.py: 'Blur - Quality: 8.514 JOD'), heat map on as in the example."""
    import colorvideovdp_amd as cv
    from conftest import kat_wavy_facade
    g, test, ref = kat_wavy_facade()
    jod, stats = cv.cvvdp(display_name="standard_4k", heatmap="threshold").predict(test, ref, dim_order="HWC")
    assert abs(float(jod) - 8.514) < 1.5e-3                      # documented, 3 decimals
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL          # the real reference on the same samples
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    hm = stats["heatmap"]
    assert tuple(hm.shape) == (1, 3, 1, 683, 1024) and hm.dtype == torch.float16
    _check_heatmap(hm[0, :, 0, ::8, ::8], g["heatmap_ds"], "kat_wavy_facade")
    # examples/ex_batch_of_images.py: the same pair inside a batch ("BHWC"), next to an identical pair (10 JOD)
    jb, sb = cv.cvvdp(display_name="standard_4k").predict(np.stack([test, ref]), np.stack([ref, ref]), dim_order="BHWC")
    assert tuple(jb.shape) == (2,) and abs(float(jb[0]) - float(g["jod"])) <= JOD_TOL and abs(float(jb[1]) - 10.0) <= 1e-6
    np.testing.assert_allclose(sb["Q_per_ch"][0:1], g["Q_per_ch"], rtol=2e-4, atol=2e-6)


def test_documented_hdr_known_answer():
    """The reference's HDR KAT (examples/ex_hdr_images.py: 'Blur - Quality: 8.696 JOD'): float32 cd/m^2 input, custom
    linear-EOTF photometry on a named display geometry."""
    import colorvideovdp_amd as cv
    from conftest import kat_nancy_church
    g, test, ref, photo = kat_nancy_church()
    disp = cv.vvdp_display_photo_eotf(photo["Y_peak"], contrast=photo["contrast"], source_colorspace=photo["source_colorspace"],
                                      EOTF=photo["EOTF"], E_ambient=photo["E_ambient"])
    jod, stats = cv.cvvdp(display_name="standard_hdr_linear", display_photometry=disp, heatmap="threshold").predict(test, ref, dim_order="HWC")
    assert abs(float(jod) - 8.696) < 1.5e-3
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)


def test_reference_video_example_blur_over_time():
    """examples/ex_blur_over_time.py through the HIP path against the real reference's outputs: 240 frames of 800x1200 at
    30 fps from host memory (16-frame blocks, upload one block ahead), mixed kernel variants down the pyramid."""
    from scipy.ndimage import gaussian_filter
    import colorvideovdp_amd as cv
    g = load_golden("kat_tree_blur_over_time")
    img, N, fps = g["img"], int(g["frames"]), float(g["fps"])
    ref = np.repeat(img[..., np.newaxis], N, axis=3)
    sig = np.concatenate((np.linspace(0.01, 2, N // 2), np.linspace(2, 0.01, N // 2)))
    test = np.zeros_like(ref)
    for f, s in enumerate(sig):
        for c in range(3):
            test[..., c, f] = gaussian_filter(ref[..., c, f], s, mode="nearest", truncate=2.0)
    met = cv.cvvdp(display_name="standard_4k")
    jod, stats = met.predict(test, ref, dim_order="HWCF", frames_per_second=fps)
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    # the same clip resident on the device (64-frame blocks, raw-halo route) gives the same features bit for bit
    jod_d, stats_d = met.predict(torch.from_numpy(test).cuda(), torch.from_numpy(ref).cuda(), dim_order="HWCF", frames_per_second=fps)
    np.testing.assert_array_equal(stats_d["Q_per_ch"], stats["Q_per_ch"])


@pytest.mark.parametrize("fps", [16, 40, 72, 100, 128])
def test_filter_lengths_without_their_own_kernel(fps):
    """Frame rates whose filter length has no register-window instantiation run on the next longer kernel with zero taps
    in front (16 -> 5 taps on the 7-tap kernel, 40 -> 11/13, 72 -> 19/25, 100 -> 27/31); 128 fps (33 taps) takes the generic
    kernel.  Against the oracle, and bit-exact over block sizes and padding modes."""
    import colorvideovdp_amd as cv
    from oracle import cvvdp_oracle as orc
    rng = np.random.default_rng(fps)
    F, H, W = 14, 48, 80
    y, x = np.mgrid[0:H, 0:W]
    ref = np.stack([np.stack([0.5 + 0.3 * np.sin(2 * np.pi * (2 * x / W + f / 9.0) + c) * np.cos(2 * np.pi * y / H) for c in range(3)]) for f in range(F)], axis=1)[None]
    test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
    ref = np.clip(ref, 0, 1).astype(np.float32)
    for pad in ("replicate", "symmetric"):
        _, ostats = orc.Oracle("standard_fhd", temp_padding=pad).predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
        qs = []
        for nb in (None, 3):
            for dev in (False, True):
                t, r = (torch.from_numpy(test).cuda(), torch.from_numpy(ref).cuda()) if dev else (test, ref)
                _, st = cv.cvvdp(display_name="standard_fhd", temp_padding=pad, block_frames=nb).predict(t, r, dim_order="BCFHW", frames_per_second=fps)
                qs.append(st["Q_per_ch"])
        np.testing.assert_allclose(qs[0], ostats["Q_per_ch"], rtol=2e-4, atol=2e-6)
        for q in qs[1:]:
            np.testing.assert_array_equal(qs[0], q)


# ---------------------------------------------------------------------------------------------------------------------
# the clips bench.py times, whole (not prefixes), against the real reference (oracle/make_goldens_bench.py)
@pytest.mark.parametrize("name", __import__("conftest").bench_cases())
def test_bench_clip_against_reference(name):
    """BASELINE.json's configurations at their own geometry: the full 1080p x 64 and 4K x 64 fp32 bench clips, the 4K x 256
    clip (configs[2], uint8) and the 8K PQ heat-map outputs (configs[4]) against outputs of the real reference on the same
    samples (regenerated from the seed on the CPU and verified by checksum)."""
    import bench
    import colorvideovdp_amd as cv
    g = load_golden(name)
    W, H, F, fps, disp, dtype = int(g["width"]), int(g["height"]), int(g["frames"]), float(g["fps"]), str(g["display"]), str(g["dtype"])
    heat = str(g["heatmap_mode"]) if "heatmap_mode" in g else None
    clip = bench.ResidentClip(F, 0, F, H, W, fps, dtype, torch.device("cuda"), gen="cpu")
    if (clip.checksum_test, clip.checksum_ref) != (int(g["checksum_test"]), int(g["checksum_ref"])):
        pytest.fail("this torch build's CPU generator does not reproduce the fixture's synthetic frames (checksum mismatch): the BASELINE-size parity check cannot run -- regenerate tests/golden with oracle/make_goldens_fullsize.py / make_goldens_bench.py")
    m = cv.cvvdp(display_name=disp, heatmap=heat)
    jod, stats = m.predict_video_source(clip)
    assert m.last_block_frames == min(240, F) or heat is not None      # the geometry the bench runs (resident raw clips: blocks up to the core's window)
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(stats["rho_band"], g["rho_band"], rtol=1e-12)
    if heat is not None:
        hm = stats["heatmap"]
        assert tuple(hm.shape) == (1, 3, F, H, W) and hm.dtype == torch.float16
        _check_heatmap(hm[0, :, :, ::16, ::16], g["heatmap_ds"], name)
        assert abs(float(hm.float().mean()) - float(g["heatmap_mean"])) < 2e-4


def test_configs2_at_full_length_in_fp32_against_reference():
    """configs[2] as bench.py runs it: 3840x2160 x 256 frames, fp32 input (the fp32 4K x 256 line had no reference
    figure).  The reference turns 8-bit codes into code / 255 in fp32 before anything else (video_source.py:320-346), so its scores for
    the fp32 clip bench.py makes (the same codes / 255) are those of the uint8 fixture `bench_4k256_u8` -- while on this side the fp32
    clip takes another route through the temporal kernel (the EOTF evaluated per sample instead of the per-code table)."""
    import bench
    import colorvideovdp_amd as cv
    g = load_golden("bench_4k256_u8")
    W, H, F, fps, disp = int(g["width"]), int(g["height"]), int(g["frames"]), float(g["fps"]), str(g["display"])
    clip = bench.ResidentClip(F, 0, F, H, W, fps, "f32", torch.device("cuda"), gen="cpu")
    if (clip.checksum_test, clip.checksum_ref) != (int(g["checksum_test"]), int(g["checksum_ref"])):
        pytest.fail("this torch build's CPU generator does not reproduce the fixture's synthetic frames (checksum mismatch)")
    m = cv.cvvdp(display_name=disp)
    jod, stats = m.predict_video_source(clip)
    assert m.last_block_frames == 240 and m.fused_levels == 3            # two temporal blocks (240 + 16 frames), the fused band kernels
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)


def test_configs4_clip_at_full_length_first_80_frames_against_reference():
    """configs[4]'s clip as bench.py makes it -- 7680x4320, PQ, uint8 codes in the PQ range -- scored at its FULL length of 256 frames
    (several temporal blocks at 8K; VERDICT r4 weak #3: nothing beyond 64 frames at 8K had met the reference).  The reference's scores
    exist for the first 80 frames (oracle/make_goldens_8k80.py: 56 minutes of its CPU path); the temporal filter is causal, so they are
    the first 80 frames' scores of the 256-frame clip."""
    import bench
    import colorvideovdp_amd as cv
    g = load_golden("deep_8k_pq_80f")
    W, H, Fg = int(g["width"]), int(g["height"]), int(g["frames"])
    F = 256
    # the first 80 frames from the CPU generator the fixture was made with, the other 176 from the (much faster) device generator: what
    # follows frame 79 cannot change the scores of frames 0..79
    clip = bench.ResidentClip(F, 0, F, H, W, float(g["fps"]), "u8", torch.device("cuda"), gen="gpu", pq_range=True)
    head = bench.ResidentClip(Fg, 0, Fg, H, W, float(g["fps"]), "u8", torch.device("cuda"), gen="cpu", pq_range=True)
    clip.test[:, :, :Fg], clip.ref[:, :, :Fg] = head.test, head.ref
    del head
    cs = (int(clip.test[:, :, :Fg].to(torch.int64).sum()), int(clip.ref[:, :, :Fg].to(torch.int64).sum()))
    if cs != (int(g["checksum_test"]), int(g["checksum_ref"])):
        pytest.fail("this torch build's CPU generator does not reproduce the fixture's synthetic frames (checksum mismatch)")
    m = cv.cvvdp(display_name=str(g["display"]))
    jod, stats = m.predict_video_source(clip)
    assert stats["Q_per_ch"].shape[2] == F and m.last_block_frames < F and m.fused_levels >= 3      # more than one temporal block
    q = stats["Q_per_ch"][:, :, :Fg]
    np.testing.assert_allclose(q, g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(stats["rho_band"], g["rho_band"], rtol=1e-12)
    jod80 = m.do_pooling_and_jods(torch.as_tensor(q, device=m.device))
    assert abs(float(jod80) - float(g["jod"])) <= JOD_TOL
    # ... and the clip cut differently (one block of 64 + ...) gives the same bits
    _release(m)                                              # (its ~100 GB workspace goes back before the next metric asks for its own)
    del m
    m2 = cv.cvvdp(display_name=str(g["display"]), block_frames=64)
    _, s2 = m2.predict_video_source(clip)
    np.testing.assert_array_equal(s2["Q_per_ch"], stats["Q_per_ch"])
    _release(m2)
    del clip, m2
    torch.cuda.empty_cache()


def test_configs4_clip_last_64_frames_against_reference():
    """The END of configs[4]'s clip: the last 64 frames (192..255 of the 256-frame clip) against the real reference's scores of the WHOLE clip
    (tests/golden/deep_8k_pq_256f.npz, oracle/make_goldens_8k256_resume.py: three hours of its CPU path, made resumably from windows of the
    causal filter; VERDICT r5 missing #2: frames 80-255 had no reference figures).  A frame's scores depend on the 16 frames before it and on nothing else (causal 17-tap filter), so only
    frames 176..255 need to be the CPU generator's (0.8 s each on the box's host cores: why the suite takes a window and
    tools/check_8k256_against_reference.py -- profiles/r06b_8k256_full_check.txt -- the whole clip); the frames before them come from the
    device generator.  The first 80 frames' entries of the fixture are the 80-frame fixture's, bit for bit (tests/test_oracle_vs_golden.py)."""
    import bench
    import colorvideovdp_amd as cv
    # the longest fixture of the clip's scores beyond the 80-frame one: deep_8k_pq_256f, or -- the resumable generator writes what it has
    # finished when it is stopped (oracle/make_goldens_8k256_resume.py N) -- a shorter prefix; the test takes its length from the fixture
    import glob
    import re
    have = [(re.search(r"deep_8k_pq_(\d+)f\.npz$", p), p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "deep_8k_pq_*f.npz"))]
    have = sorted((int(m.group(1)), p) for m, p in have if m is not None and int(m.group(1)) > 80)     # (not deep_8k_pq_heat_17f: another clip)
    if not have:
        pytest.skip("no tests/golden/deep_8k_pq_<N>f.npz beyond 80 frames (oracle/make_goldens_8k256_resume.py)")
    g = load_golden(os.path.basename(have[-1][1])[:-4])
    W, H, F = int(g["width"]), int(g["height"]), int(g["frames"])
    lo, first = F - 80, F - 64
    clip = bench.ResidentClip(F, 0, F, H, W, float(g["fps"]), "u8", torch.device("cuda"), gen="gpu", pq_range=True)
    tail = bench.ResidentClip(F, lo, F, H, W, float(g["fps"]), "u8", torch.device("cuda"), gen="cpu", pq_range=True)
    clip.test[:, :, lo:], clip.ref[:, :, lo:] = tail.test, tail.ref
    del tail
    m = cv.cvvdp(display_name=str(g["display"]))
    jod, stats = m.predict_video_source(clip)
    assert stats["Q_per_ch"].shape[2] == F and m.last_block_frames < F
    np.testing.assert_allclose(stats["Q_per_ch"][:, :, first:], g["Q_per_ch"][:, :, first:], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(stats["rho_band"], g["rho_band"], rtol=1e-12)
    _release(m)
    del clip, m
    torch.cuda.empty_cache()


def test_configs4_as_stated_256_frames_with_heat_map_and_distogram():
    """configs[4] AS BASELINE.json STATES IT -- 7680x4320, PQ, 256 frames, supra-threshold heat map + distogram -- on the clip bench.py
    --workload 8k256pq times (codes in the PQ range), the heat-map frames consumed on the GPU in pieces of a long temporal block
    (VERDICT r5 next #3).  What the real reference says about this clip:
      * Q_per_ch of frames 0..79      tests/golden/deep_8k_pq_80f.npz            (oracle/make_goldens_8k80.py)
      * heat-map frames 0, 8, 16, the per-frame means of frames 0..16 and the distogram arrays of the 17-frame prefix
                                      tests/golden/deep_8k_pqrange_heat_17f.npz  (oracle/make_goldens_8k17_pqrange.py)
    (the temporal filter is causal and the tone curve per frame: a prefix's outputs are the long clip's).  And the same bits whatever
    the cut: 16- against 32-frame pieces, and a shorter temporal block."""
    import bench
    import colorvideovdp_amd as cv
    g80, g17 = load_golden("deep_8k_pq_80f"), load_golden("deep_8k_pqrange_heat_17f")
    W, H, F80, F17, F = int(g80["width"]), int(g80["height"]), int(g80["frames"]), int(g17["frames"]), 256
    fps, disp = float(g80["fps"]), str(g80["display"])
    # frames 0..79 from the CPU generator the fixtures were made with (shared with the other 8K tests), 80..255 from the device generator
    clip = bench.ResidentClip(F, 0, F, H, W, fps, "u8", torch.device("cuda"), gen="gpu", pq_range=True)
    head = bench.ResidentClip(F80, 0, F80, H, W, fps, "u8", torch.device("cuda"), gen="cpu", pq_range=True)
    clip.test[:, :, :F80], clip.ref[:, :, :F80] = head.test, head.ref
    del head
    cs = lambda n: (int(clip.test[:, :, :n].sum(dtype=torch.int64)), int(clip.ref[:, :, :n].sum(dtype=torch.int64)))     # noqa: E731
    if cs(F80) != (int(g80["checksum_test"]), int(g80["checksum_ref"])) or cs(F17) != (int(g17["checksum_test"]), int(g17["checksum_ref"])):
        pytest.fail("this torch build's CPU generator does not reproduce the fixtures' synthetic frames (checksum mismatch)")
    keep = [int(k) for k in g17["heatmap_frames"]]
    # round 6: the reference's heat map of the first 64 frames (oracle/make_goldens_8k64_heat.py: into the second temporal block of this run,
    # across three 16-frame pieces) -- the mean of every frame and six frames down-sampled; VERDICT r5 missing #2
    g64 = load_golden("deep_8k_pqrange_heat_64f") if os.path.isfile(os.path.join(os.path.dirname(__file__), "golden", "deep_8k_pqrange_heat_64f.npz")) else None
    keep64 = [int(k) for k in g64["heatmap_frames"]] if g64 is not None else []
    ds64 = int(g64["heatmap_ds_step"]) if g64 is not None else 1

    class Sink:                          # what a consumer on the GPU sees: every piece as a device tensor
        wants_device, wants_uint8 = True, False

        def __init__(self):
            self.firsts, self.kept, self.kept64, self.means = [], {}, {}, []

        def __call__(self, first, frames):
            assert frames.is_cuda and frames.dtype == torch.float16 and tuple(frames.shape[:2]) == (1, 3) and tuple(frames.shape[3:]) == (H, W)
            self.firsts.append((first, frames.shape[2]))
            # (frame by frame: one reduction shape whatever the piece length, so that the means of two cuts can be compared bit for bit)
            self.means.append(torch.stack([frames[0, :, i].float().mean() for i in range(frames.shape[2])]))
            for k in keep:
                if first <= k < first + frames.shape[2]:
                    self.kept[k] = frames[0, :, k - first, ::16, ::16].clone()
            for k in keep64:
                if first <= k < first + frames.shape[2]:
                    self.kept64[k] = frames[0, :, k - first, ::ds64, ::ds64].clone()

    runs = []
    for score_frames, block in ((None, None), (32, None), (None, 40)):
        m = cv.cvvdp(display_name=disp, heatmap=str(g17["heatmap_mode"]), block_frames=block)
        if score_frames is not None:
            m.score_frames = score_frames
        sink = Sink()
        jod, stats = m.predict_video_source(clip, heatmap_sink=sink)
        assert "heatmap" not in stats and stats["Q_per_ch"].shape[2] == F
        pos = 0
        for first, n in sink.firsts:                                       # the pieces tile the clip in order
            assert first == pos
            pos += n
        assert pos == F and len(sink.firsts) >= F // 32
        assert m.last_block_frames < F                                     # more than one temporal block
        runs.append((stats["Q_per_ch"], torch.stack([sink.kept[k] for k in keep], dim=1).cpu(), torch.cat(sink.means).cpu().numpy(), float(jod),
                     {k: v for k, v in stats.items() if k != "heatmap"}))
        if len(runs) == 1 and g64 is not None:
            assert m.last_block_frames < int(g64["frames"])               # the fixture's last frames belong to this run's SECOND temporal block
            hm_keep64 = torch.stack([sink.kept64[k] for k in keep64], dim=1).cpu()
        _release(m)                                                        # (one metric's workspace at a time: tens of gigabytes at 8K)
        del m, sink
    q, hm_keep, means, jod, stats = runs[0]
    m = cv.cvvdp(display_name=disp)                                        # (pooling / distogram arithmetic only: no workspace)
    # ---- the reference: scores of frames 0..79, heat-map frames 0 / 8 / 16, the means of frames 0..16, the distogram of the 17-frame prefix
    np.testing.assert_allclose(q[:, :, :F80], g80["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(q[:, :, :F17], g17["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(stats["rho_band"], g80["rho_band"], rtol=1e-12)
    assert abs(float(m.do_pooling_and_jods(torch.as_tensor(q[:, :, :F80], device=m.device))) - float(g80["jod"])) <= JOD_TOL
    assert abs(float(m.do_pooling_and_jods(torch.as_tensor(q[:, :, :F17], device=m.device))) - float(g17["jod"])) <= JOD_TOL
    _check_heatmap(hm_keep, g17["heatmap_ds"], "deep_8k_pqrange_heat_17f")
    np.testing.assert_allclose(means[:F17], g17["heatmap_frame_means"], atol=2e-4)
    if g64 is not None:
        F64 = int(g64["frames"])
        np.testing.assert_allclose(q[:, :, :F64], g64["Q_per_ch"], rtol=2e-4, atol=2e-6)
        _check_heatmap(hm_keep64, g64["heatmap_ds"], "deep_8k_pqrange_heat_64f")
        np.testing.assert_allclose(means[:F64], g64["heatmap_frame_means"], atol=2e-4)
    st17 = dict(stats, Q_per_ch=q[:, :, :F17], N_frames=F17)
    for jm, key in ((None, "disto_auto"), (10, "disto_10")):               # cvvdp_metric.py:1160-1192: what imshow is handed
        panels, _ = m.distogram_data(st17, jod_max=jm)
        assert panels.shape == g17[key].shape
        np.testing.assert_allclose(panels, g17[key], rtol=5e-4, atol=2e-6)
    panels, _ = m.distogram_data(stats, jod_max=10)                         # ... and the 256-frame distogram exists and is finite
    assert panels.shape[-1] == F and np.isfinite(panels).all()
    # ---- cut differently: the same bits (scores, heat-map frames, means) and the same JOD
    for q2, hm2, means2, jod2, _ in runs[1:]:
        np.testing.assert_array_equal(q2, q)
        assert torch.equal(hm2, hm_keep)
        np.testing.assert_array_equal(means2, means)
        assert jod2 == jod
    del clip, runs
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------
# host-side outputs and the rest of the class API against the real reference (oracle/make_goldens_outputs.py)
def _outputs():
    return load_golden("outputs")


@pytest.mark.parametrize("tag", ["vid", "img"])
def test_distogram_and_features_against_reference(tag, tmp_path):
    import json
    o = _outputs()
    g = load_golden(str(o[f"{tag}_case"]))
    meta = g["meta"]
    m = _metric(dict(meta, heatmap=None))
    jod, stats = m.predict(*_inputs(g), dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    for jm, key in ((None, f"{tag}_disto_auto"), (10, f"{tag}_disto_10")):
        panels, _ = m.distogram_data(stats, jod_max=jm)
        assert panels.shape == o[key].shape
        np.testing.assert_allclose(panels, o[key], rtol=5e-4, atol=2e-6)       # cvvdp_metric.py:1160-1192: what imshow is handed
        png = tmp_path / f"d_{jm}.png"
        m.export_distogram(stats, str(png), jod_max=jm)
        assert png.stat().st_size > 2000 and png.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"
    m.write_features_to_json(stats, str(tmp_path / "f.json"))                # :1112-1127
    got, want = json.load(open(tmp_path / "f.json")), json.loads(str(o[f"{tag}_features_json"]))
    assert list(got.keys()) == list(want.keys())
    for k in want:
        if isinstance(want[k], list):
            np.testing.assert_allclose(np.asarray(got[k], dtype=np.float64), np.asarray(want[k], dtype=np.float64), rtol=2e-4, atol=2e-6)
        else:
            assert got[k] == want[k], k


def test_loss_against_reference():
    """loss() = 10 - JOD (cvvdp_metric.py:294-298); inference only: gradients are refused, not silently dropped."""
    import colorvideovdp_amd as cv
    o = _outputs()
    g = load_golden(str(o["vid_case"]))
    meta = g["meta"]
    m = _metric(dict(meta, heatmap=None))
    loss = m.loss(*_inputs(g), dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    assert abs(float(loss) - float(o["vid_loss"])) <= JOD_TOL
    x = torch.rand((3, 32, 32), requires_grad=True)
    with pytest.raises(cv.vq_exception):
        m.loss(x, x.detach(), dim_order="CHW")


def test_temporally_filtered_source_bypasses_the_fir():
    """vid_source.is_temporally_filtered (cvvdp_metric.py:470-488): 4-channel 'DKLd65_trans' frames go straight into the
    level-0 planes.  Fed with the planes the reference filtered itself, the result must match the reference's; fed with this
    build's own filtered planes it must reproduce the normal path bit for bit."""
    from colorvideovdp_amd import _capi
    o = _outputs()
    g = load_golden(str(o["vid_case"]))
    meta = dict(g["meta"], heatmap=None)
    T, R = torch.from_numpy(o["prefiltered_test"]), torch.from_numpy(o["prefiltered_ref"])     # [4, F, H, W]

    class Pre:
        is_temporally_filtered = True
        calls = []

        def __init__(self, T, R):
            self.T, self.R = T, R

        def get_video_size(self):
            return (self.T.shape[2], self.T.shape[3], self.T.shape[1])

        def get_frames_per_second(self):
            return meta["fps"]

        def get_batch_size(self):
            return 1

        def get_test_frame(self, f, device, colorspace):
            assert colorspace == "DKLd65_trans"
            Pre.calls.append(("t", f))
            return self.T[None, :, f:f + 1].to(device)

        def get_reference_frame(self, f, device, colorspace):
            assert colorspace == "DKLd65_trans"
            Pre.calls.append(("r", f))
            return self.R[None, :, f:f + 1].to(device)

    m = _metric(meta, block_frames=5)
    jod, stats = m.predict_video_source(Pre(T, R))
    np.testing.assert_allclose(stats["Q_per_ch"], o["prefiltered_Q_per_ch"], rtol=2e-4, atol=2e-6)
    assert abs(float(jod) - float(o["prefiltered_jod"])) <= JOD_TOL
    F = T.shape[1]
    assert Pre.calls == [x for f in range(F) for x in (("r", f), ("t", f))]      # each frame once, in order, reference first
    # own planes: run the normal path with the level-0 planes kept, feed them back
    j_n, s_n = _metric(meta).predict(*_inputs(g), dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    m2 = _metric(meta)            # (a second run keeps the buffers; its band kernels are the per-pixel-dump instantiation, which
    m2.debug_dump = True          # may round a last bit differently: the level-0 planes come from the same temporal kernel)
    m2.predict(*_inputs(g), dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    H, W = T.shape[2], T.shape[3]
    planes = m2.debug_buffer(_capi.BUF_GPYR, 0)[:8 * F * H * W].view(8, F, H, W).cpu()
    j_p, s_p = _metric(meta).predict_video_source(Pre(planes[0::2].contiguous(), planes[1::2].contiguous()))
    np.testing.assert_array_equal(s_p["Q_per_ch"], s_n["Q_per_ch"])
    assert float(j_p) == float(j_n)


def test_generic_source_is_read_once_and_in_order():
    """the reference's file sources are strictly sequential.  A generic video_source (DKL frames, one by one)
    must see every frame exactly once, in increasing order -- also with symmetric padding (read-ahead) and small blocks --
    and a display-photometry object of another package on the source must not matter."""
    from oracle import cvvdp_oracle as orc
    g = load_golden("vid_u16_67x121x20_30_4k_sym")
    meta = g["meta"]
    assert meta["temp_padding"] == "symmetric"
    disp = orc.Display(meta["display"])
    t, r = _inputs(g)
    tt, rr = orc.to_bcfhw(t, meta["dim_order"]), orc.to_bcfhw(r, meta["dim_order"])
    F = tt.shape[2]

    class Foreign:                       # stands in for pycvvdp.vvdp_display_photo_eotf
        pass

    class Src:
        dm_photometry = Foreign()

        def __init__(self):
            self.log = {"t": [], "r": []}

        def get_video_size(self):
            return (tt.shape[3], tt.shape[4], F)

        def get_batch_size(self):
            return 1

        def get_frames_per_second(self):
            return meta["fps"]

        def _frame(self, which, src, frame, colorspace):
            assert colorspace == "DKLd65"
            assert frame == (self.log[which][-1] + 1 if self.log[which] else 0), f"random access: {which} {frame} after {self.log[which]}"
            self.log[which].append(frame)
            return disp.to_dkl(orc.fetch_frame(src, frame))

        def get_test_frame(self, frame, device, colorspace):
            return self._frame("t", tt, frame, colorspace).to(device)

        def get_reference_frame(self, frame, device, colorspace):
            return self._frame("r", rr, frame, colorspace).to(device)

    for nb in (None, 4):
        src = Src()
        jod, stats = _metric(meta, block_frames=nb).predict_video_source(src)
        assert src.log["t"] == list(range(F)) and src.log["r"] == list(range(F))
        np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(jod.cpu().numpy(), g["jod"], atol=JOD_TOL)


def test_streaming_heatmap_sink_matches_the_whole_clip_tensor(tmp_path):
    """with heatmap_sink the frames arrive block by block (bounded host memory) and are the same fp16 values
    stats["heatmap"] would hold; the writers put them on disk (PNG sequence, .npy in the reference's layout)."""
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import heatmap_writers as hw
    g = load_golden("vid_u8_135x240x18_60_fhd_raw")
    meta = dict(g["meta"])
    t, r = _inputs(g)
    kw = dict(dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    for mode in ("raw", "supra-threshold"):
        meta["heatmap"] = mode
        _, s_full = _metric(meta).predict(t, r, **kw)
        full = s_full["heatmap"].clone()
        F, H, W = full.shape[2], full.shape[3], full.shape[4]
        got = torch.zeros_like(full)
        order = []

        def sink(first, frames):
            order.append((first, frames.shape[2]))
            got[:, :, first:first + frames.shape[2]] = frames

        m = _metric(meta, block_frames=5)
        vs = cv.video_source_array(t, r, meta["fps"], dim_order=meta["dim_order"], display_photometry=m.display_photometry)
        jod, stats = m.predict_video_source(vs, heatmap_sink=sink)
        assert "heatmap" not in stats
        assert order == [(f, min(5, F - f)) for f in range(0, F, 5)]
        assert torch.equal(got, full)                                      # per-frame statistics: blocking does not matter
        np.testing.assert_array_equal(stats["Q_per_ch"], s_full["Q_per_ch"])
    npy = hw.HeatmapNpyWriter(str(tmp_path / "hm.npy"), F, H, W, channels=3)
    png = hw.HeatmapPngWriter(str(tmp_path / "seq" / "hm_%03d.png"))
    m.predict_video_source(vs, heatmap_sink=lambda f, x: (npy(f, x), png(f, x)))
    npy.close()
    assert np.array_equal(np.load(tmp_path / "hm.npy"), full.numpy())
    assert png.frames_written == F and (tmp_path / "seq" / ("hm_%03d.png" % (F - 1))).stat().st_size > 500
    with pytest.raises(cv.vq_exception):
        _metric(dict(meta, heatmap=None)).predict_video_source(vs, heatmap_sink=sink)


def test_heat_colour_kernels_agree_bit_for_bit():
    """The heat-map finishing kernel of large frames (heatmap.hip k_heat_colour_rows: a block owns a tile of rows, tone curve and colour map in
    LDS, the coarse patch of the fused last reconstruction step as a rolling window) against the kernel it replaced on those frames
    (k_heat_colour<true>: a thread = 4 pixels anywhere; still what W % 4 == 0 frames without a fused last step run, and band_layout = 1
    selects it everywhere): the same operations on the same values, so the same fp16 / 8-bit frames bit for bit.  fuse_mode = 2 keeps the
    band kernels out of the comparison (band_layout also switches the fused levels' band kernel).  Shapes: two column chunks with a
    part-filled last one (W = 1028), an odd number of rows and a last row tile of 7 rows; a tone curve from the histogram and the
    linear one of a frame with less than 0.6 log units of range (visualize_diff_map.py:28-31).  The same switch selects k_expand_add4
    where the coarser steps of the reconstruction run its row-tile form (pyramid.hip k_expand_add_rows: levels at least 512 columns wide
    and 32 rows high with W % 4 == 0 -- levels 1 of the 2056- and 1, 2 of the 4096-column frames here)."""
    import colorvideovdp_amd as cv
    rng = np.random.default_rng(77)
    for (H, W, F), flat in (((71, 1028, 3), False), ((180, 256, 4), False), ((48, 64, 2), True), ((130, 2056, 2), False), ((140, 4096, 2), False)):
        yy, xx = np.mgrid[0:H, 0:W]
        if flat:
            base = 118.0 + 6.0 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
        else:
            base = 8.0 + 235.0 * (0.5 + 0.5 * np.sin(xx / 37.0 + yy / 23.0)) * (xx / W)
        r = np.clip(base[None, None, :, :] + rng.normal(0, 2.0, (F, 3, H, W)), 0, 255).round().astype(np.uint8)        # [F,C,H,W]
        t = np.clip(r.astype(np.float32) + rng.normal(0, 6.0, r.shape) * (xx > W // 3), 0, 255).round().astype(np.uint8)
        t, r = (torch.from_numpy(np.ascontiguousarray(a.transpose(1, 0, 2, 3))[None]) for a in (t, r))                  # BCFHW
        for mode in ("threshold", "supra-threshold", "raw"):                 # (raw: k_heat_raw_rows against k_heat_raw4)
            out = {}
            for layout in (0, 1):
                m = cv.cvvdp(display_name="standard_4k", heatmap=mode)
                m.fuse_mode, m.band_layout = 2, layout
                _, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=30)
                got = np.zeros((F, H, W, 1 if mode == "raw" else 3), np.uint8)

                class Sink:
                    wants_uint8 = True

                    def __call__(self, first, frames):
                        got[first:first + frames.shape[0]] = frames.numpy()

                vs = cv.video_source_array(t, r, 30, dim_order="BCFHW", display_photometry=m.display_photometry)
                m.predict_video_source(vs, heatmap_sink=Sink())
                out[layout] = (st["heatmap"].numpy().copy(), st["Q_per_ch"].copy(), got)
            np.testing.assert_array_equal(out[0][1], out[1][1])
            a16, b16 = out[0][0].view(np.uint16).astype(np.int32), out[1][0].view(np.uint16).astype(np.int32)
            bad = np.argwhere(a16 != b16)
            assert bad.shape[0] == 0, (H, W, mode, "fp16 planes differ", bad.shape[0], int(np.abs(a16 - b16).max()), bad[:8].tolist())
            np.testing.assert_array_equal(out[0][2], out[1][2])
            assert out[0][0].astype(np.float32).std() > 0.01 and np.isfinite(out[0][0].astype(np.float32)).all()


def test_uint8_heatmap_sink_is_the_writers_conversion(tmp_path):
    """A sink with wants_uint8 receives [n, H, W, C] uint8 frames made on the GPU (cvvdp_get_heatmap_rgb8): bit for bit what the
    reference's writers make of the fp16 map (np.clip(x, 0, 1) * 255 -> uint8, run_cvvdp.py:62-78); half the bytes over PCIe."""
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import heatmap_writers as hw
    from PIL import Image
    g = load_golden("vid_u8_135x240x18_60_fhd_raw")
    meta = dict(g["meta"])
    t, r = _inputs(g)
    for mode in ("raw", "threshold", "supra-threshold"):
        meta["heatmap"] = mode
        m = _metric(meta, block_frames=7)
        _, s_full = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
        want = hw.heatmap_to_uint8(s_full["heatmap"])                      # the host conversion of the fp16 tensor: [F, H, W, 3]
        # ... which must be the reference writers' own expression on the float16 array (run_cvvdp.py:59-63, :76): the product
        # stays float16 (rounded to half) before the truncation
        lit = s_full["heatmap"][0].permute(1, 2, 3, 0).numpy()
        assert lit.dtype == np.float16
        lit = (np.clip(lit, 0.0, 1.0) * 255.0).astype(np.uint8)
        np.testing.assert_array_equal(want, np.concatenate([lit] * 3, -1) if lit.shape[-1] == 1 else lit)
        C = 1 if mode == "raw" else 3
        got = np.zeros(want.shape[:3] + (C,), np.uint8)

        class Sink:
            wants_uint8 = True

            def __call__(self, first, frames):
                assert frames.dtype == torch.uint8 and frames.shape[1:] == got.shape[1:]
                got[first:first + frames.shape[0]] = frames.numpy()

        vs = cv.video_source_array(t, r, meta["fps"], dim_order=meta["dim_order"], display_photometry=m.display_photometry)
        _, stats = m.predict_video_source(vs, heatmap_sink=Sink())
        assert "heatmap" not in stats
        np.testing.assert_array_equal(np.concatenate([got] * 3, -1) if C == 1 else got, want)
        np.testing.assert_array_equal(stats["Q_per_ch"], s_full["Q_per_ch"])
    png = hw.HeatmapPngWriter(str(tmp_path / "hm_%02d.png"))
    m.predict_video_source(vs, heatmap_sink=png)
    assert png.frames_written == want.shape[0]
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "hm_11.png")), want[11])
    means8, means16 = hw.HeatmapFrameMeans(step=4, uint8=True), hw.HeatmapFrameMeans(step=4)
    m.predict_video_source(vs, heatmap_sink=means8)
    m.predict_video_source(vs, heatmap_sink=means16)
    assert means8.frames_seen == means16.frames_seen == want.shape[0]
    assert max(np.abs(means8.means[f] - means16.means[f]).max() for f in means8.means) < 1.0 / 255


@pytest.mark.parametrize("W", [241, 242, 243, 244, 245, 246, 247, 248, 249, 250, 251, 252, 253, 254, 255, 341, 342, 483, 484, 683])
def test_ragged_widths_against_oracle(W):
    """every width takes the marching reduce and the fused band kernel (W % 8 in 1..7, odd and even, the
    last lane of the last strip holding 1..3 valid columns, one and several strips, odd heights for the row-parity quirk of
    lpyr_dec.py:206) -- features against the oracle, per-pixel D of every level against the oracle for one of them."""
    import colorvideovdp_amd as cv
    from oracle import cvvdp_oracle as orc
    rng = np.random.default_rng(W)
    H = 97 if W % 2 else 80
    F, fps = (3, 30) if W < 300 else (2, 60)
    y, x = np.mgrid[0:H, 0:W]
    ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (4.3 * x / W + f / 7.0) + c) * np.cos(2 * np.pi * 2.7 * y / H) for c in range(3)])
                    for f in range(F)], axis=1)[None]
    test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1)        # noise right up to the right edge
    test, ref = np.round(test * 255).astype(np.uint8), np.round(np.clip(ref, 0, 1) * 255).astype(np.uint8)
    o = orc.Oracle(display_name="standard_fhd")
    ojod, ostats = o.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    m = cv.cvvdp(display_name="standard_fhd")
    jod, stats = m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    assert abs(float(jod) - float(ojod)) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], ostats["Q_per_ch"], rtol=2e-4, atol=2e-6)
    # heat map path of the ragged kernels (per-pixel outputs reach the right edge)
    oh = orc.Oracle(display_name="standard_fhd", heatmap="raw")
    _, ohs = oh.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    _, hs = cv.cvvdp(display_name="standard_fhd", heatmap="raw").predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
    _check_heatmap(hs["heatmap"], ohs["heatmap"].numpy() if torch.is_tensor(ohs["heatmap"]) else ohs["heatmap"])


def test_ml_head_features_against_reference():
    """cvvdp.extract_features() = the pooled |T_f|*S, |R_f|*S, D statistics the reference's ML heads consume
    (cvvdp_ml_metric.py:77-107, :302-390), against the real reference (oracle/make_goldens_features.py), video and image."""
    gf = load_golden("features")
    k = 0
    while f"case{k}" in gf:
        g = load_golden(str(gf[f"case{k}"]))
        meta = dict(g["meta"], heatmap=None)
        import colorvideovdp_amd as cv
        m = _metric(meta, block_frames=7)
        t, r = _inputs(g)
        vs = cv.video_source_array(t, r, meta["fps"], dim_order=meta["dim_order"], display_photometry=m.display_photometry)
        feats, hm = m.extract_features(vs)
        assert hm is None and len(feats) == int(gf[f"case{k}_bands"])
        for bb, f in enumerate(feats):
            want = gf[f"case{k}_band{bb}"]
            got = f.cpu().numpy()
            assert got.shape == want.shape, (bb, got.shape, want.shape)
            # means to the usual feature tolerance; a variance E[x^2] - mean^2 inherits the rounding of both terms
            for q in (0, 2, 4):
                np.testing.assert_allclose(got[..., q], want[..., q], rtol=5e-4, atol=2e-6, err_msg=f"band {bb} mean {q}")
                scale = np.abs(want[..., q]) ** 2 + np.abs(want[..., q + 1])
                assert np.all(np.abs(got[..., q + 1] - want[..., q + 1]) <= 2e-3 * scale + 1e-7), f"band {bb} var {q + 1}"
        # the normal path still works on the same object afterwards (features mode is per call)
        jod, stats = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
        np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
        k += 1
    assert k == 2


def test_ml_head_features_of_a_4k_clip_against_reference():
    """The FEAT instantiations of the fused band kernels at the size they are chosen for: 16 frames of the 3840x2160 bench clip, all bands'
    pooled statistics against the real reference's extract_features (oracle/make_goldens_features_4k.py)."""
    import bench
    import colorvideovdp_amd as cv
    g = load_golden("features_4k16")
    W, H, F = int(g["width"]), int(g["height"]), int(g["frames"])
    clip = bench.ResidentClip(F, 0, F, H, W, float(g["fps"]), "u8", torch.device("cuda"), gen="cpu")
    if (clip.checksum_test, clip.checksum_ref) != (int(g["checksum_test"]), int(g["checksum_ref"])):
        pytest.fail("this torch build's CPU generator does not reproduce the fixture's synthetic frames (checksum mismatch)")
    m = cv.cvvdp(display_name=str(g["display"]))
    feats, hm = m.extract_features(clip)
    assert hm is None and len(feats) == int(g["bands"]) and m.fused_levels == 2
    for bb, f in enumerate(feats):
        want, got = g[f"band{bb}"], f.cpu().numpy()
        assert got.shape == want.shape, (bb, got.shape, want.shape)
        for q in (0, 2, 4):
            # (atol: the transient channel's coarse-band means are ~1e-4, sums of a few dozen rounding-level terms; 2.1e-6 observed)
            np.testing.assert_allclose(got[..., q], want[..., q], rtol=5e-4, atol=4e-6, err_msg=f"band {bb} mean {q}")
            scale = np.abs(want[..., q]) ** 2 + np.abs(want[..., q + 1])
            assert np.all(np.abs(got[..., q + 1] - want[..., q + 1]) <= 2e-3 * scale + 1e-7), f"band {bb} var {q + 1}"


def test_large_ragged_frames_split_off_their_edge_strips():
    """W % 8 != 0 on a frame large enough that the aligned strips alone are two GPU-fulls of workgroups: those strips run the
    aligned instantiation of k_band4 and only the edge strip the RAGGED one, as two launches (core.cpp: split_edge, decided from the
    clip's nominal block).  The temporal filter is causal, so the first 8 frames of the 40-frame clip (split) must score like the
    8-frame clip (one RAGGED launch: the path the small ragged shapes are checked on against the oracle) -- different row segments
    and kernels, same numbers to rounding."""
    import bench
    import colorvideovdp_amd as cv
    H, W = 1444, 2566
    long = bench.ResidentClip(40, 0, 40, H, W, 60, "u8", torch.device("cuda"))
    short = bench.ResidentClip(8, 0, 8, H, W, 60, "u8", torch.device("cuda"))
    assert torch.equal(long.test[:, :, :8], short.test)
    m = cv.cvvdp(display_name="standard_4k")
    _, s_long = m.predict_video_source(long)
    fused_long = m.fused_levels
    _, s_short = m.predict_video_source(short)
    np.testing.assert_allclose(s_long["Q_per_ch"][:, :, :8], s_short["Q_per_ch"], rtol=2e-5, atol=1e-7)
    # and blocks of the long clip do not change a bit (the split is a property of the clip, not of the block)
    _, s_blk = cv.cvvdp(display_name="standard_4k", block_frames=9).predict_video_source(long)
    np.testing.assert_array_equal(s_blk["Q_per_ch"], s_long["Q_per_ch"])
    # Since the fused band kernels take even widths (W % 4 == 2 here: k_band4f<4, 2> on the border strips), level 0 of this clip
    # is fused by the product's own choice; the unfused route (reduce pass + k_band4 with its edge strips split off) must agree.
    assert fused_long == 1                           # level 1 is 1283 columns wide: odd, reduce pass
    m2 = cv.cvvdp(display_name="standard_4k")
    m2.fuse_mode = 2
    _, s_unfused = m2.predict_video_source(long)
    assert m2.fused_levels == 0
    np.testing.assert_allclose(s_long["Q_per_ch"], s_unfused["Q_per_ch"], rtol=2e-4, atol=2e-6)      # (two routes, two roundings: the generic tolerance)


def test_8k_pq_full_temporal_window_heatmap_and_distogram_against_reference():
    """configs[4] at depth (missing #4): 17 frames of the 7680x4320 PQ clip, so that frame 16 is scored with 16 DISTINCT
    real predecessors (the 2-frame 8K fixtures only see replicate-padded windows), with the supra-threshold heat map and the
    distogram arrays, all against the real reference (oracle/make_goldens_8k17.py: 24 minutes of the reference's CPU path)."""
    import bench
    import colorvideovdp_amd as cv
    g = load_golden("deep_8k_pq_heat_17f")
    W, H, F = int(g["width"]), int(g["height"]), int(g["frames"])
    clip = bench.ResidentClip(F, 0, F, H, W, float(g["fps"]), "u8", torch.device("cuda"), gen="cpu")
    if (clip.checksum_test, clip.checksum_ref) != (int(g["checksum_test"]), int(g["checksum_ref"])):
        pytest.fail("this torch build's CPU generator does not reproduce the fixture's synthetic frames (checksum mismatch)")
    m = cv.cvvdp(display_name=str(g["display"]), heatmap=str(g["heatmap_mode"]))
    jod, stats = m.predict_video_source(clip)
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(stats["Q_per_ch"][:, :, 16], g["Q_per_ch"][:, :, 16], rtol=2e-4, atol=2e-6)   # the frame with the full window
    hm = stats["heatmap"]
    assert tuple(hm.shape) == (1, 3, F, H, W) and hm.dtype == torch.float16
    keep = [int(k) for k in g["heatmap_frames"]]
    _check_heatmap(hm[0][:, keep, ::16, ::16], g["heatmap_ds"], "deep_8k_pq_heat_17f")
    means = np.array([float(hm[0, :, f].float().mean()) for f in range(F)], dtype=np.float32)
    np.testing.assert_allclose(means, g["heatmap_frame_means"], atol=2e-4)
    st = {k: v for k, v in stats.items() if k != "heatmap"}
    for jm, key in ((None, "disto_auto"), (10, "disto_10")):               # cvvdp_metric.py:1160-1192: what imshow is handed
        panels, _ = m.distogram_data(st, jod_max=jm)
        assert panels.shape == g[key].shape
        np.testing.assert_allclose(panels, g[key], rtol=5e-4, atol=2e-6)


def test_4k_heatmap_clip_in_pieces_on_the_fused_kernels_against_reference():
    """Everything round 4 added to the heat-map path at the size where all of it is active, against the real reference
    (oracle/make_goldens_4k_heat.py): 20 frames of the 3840x2160 clip resident in HBM, threshold heat map -- a 20-frame temporal block scored
    in pieces of 16 + 4 frames, two levels on k_band4s_heat / k_band4f_heat, the context plane's range from the level-0 band kernels, the
    last reconstruction step inside the finishing kernels."""
    import bench
    import colorvideovdp_amd as cv
    g = load_golden("deep_4k_thr_heat_20f")
    W, H, F = int(g["width"]), int(g["height"]), int(g["frames"])
    clip = bench.ResidentClip(F, 0, F, H, W, float(g["fps"]), "u8", torch.device("cuda"), gen="cpu")
    if (clip.checksum_test, clip.checksum_ref) != (int(g["checksum_test"]), int(g["checksum_ref"])):
        pytest.fail("this torch build's CPU generator does not reproduce the fixture's synthetic frames (checksum mismatch)")
    m = cv.cvvdp(display_name=str(g["display"]), heatmap=str(g["heatmap_mode"]))
    jod, stats = m.predict_video_source(clip)
    assert (m.last_block_frames, m.last_score_frames) == (20, 16) and m.fused_levels == 2     # (a 20-frame clip: level 2 holds 10 M pixels, below the fuse rule)
    assert abs(float(jod) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(stats["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    hm = stats["heatmap"]
    assert tuple(hm.shape) == (1, 3, F, H, W) and hm.dtype == torch.float16
    keep = [int(k) for k in g["heatmap_frames"]]
    _check_heatmap(hm[0][:, keep, ::16, ::16], g["heatmap_ds"], "deep_4k_thr_heat_20f")
    means = np.array([float(hm[0, :, f].float().mean()) for f in range(F)], dtype=np.float32)
    np.testing.assert_allclose(means, g["heatmap_frame_means"], atol=2e-4)
    st = {k: v for k, v in stats.items() if k != "heatmap"}
    for jm, key in ((None, "disto_auto"), (10, "disto_10")):
        panels, _ = m.distogram_data(st, jod_max=jm)
        assert panels.shape == g[key].shape
        np.testing.assert_allclose(panels, g[key], rtol=5e-4, atol=2e-6)
    # and the same frames through a device-resident sink, unfused, in other pieces: the same scores to rounding, the same map to fp16 codes
    m2 = cv.cvvdp(display_name=str(g["display"]), heatmap=str(g["heatmap_mode"]), block_frames=13)
    m2.fuse_mode, m2.score_frames = 2, 5
    got = torch.zeros((3, F, H // 16 + (H % 16 > 0), W // 16), dtype=torch.float16)

    class Sink:
        wants_device = True

        def __call__(self, first, frames):
            got[:, first:first + frames.shape[2]] = frames[0, :, :, ::16, ::16].cpu()

    jod2, st2 = m2.predict_video_source(clip, heatmap_sink=Sink())
    assert m2.fused_levels == 0 and abs(float(jod2) - float(g["jod"])) <= JOD_TOL
    np.testing.assert_allclose(st2["Q_per_ch"], g["Q_per_ch"], rtol=2e-4, atol=2e-6)
    _check_heatmap(got[:, keep], g["heatmap_ds"])


def test_kernel_timings_in_stats_when_asked_for():
    """HIP-event timings of the kernel families exposed in `stats` (opt-in: the reference's stats keys stay as they are)."""
    g = load_golden("vid_u8_72x128x12_60_fhd")
    meta = dict(g["meta"], heatmap=None)
    m = _metric(meta)
    t, r = _inputs(g)
    _, s0 = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    assert "kernel_ms" not in s0
    m.profile(True, per_call=True)
    _, s1 = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    km = s1["kernel_ms"]
    assert set(km) >= {"temporal_fir", "pyr_reduce", "band_level0", "band_rest"} and km["temporal_fir"] > 0 and km["band_level0"] > 0
    np.testing.assert_array_equal(s1["Q_per_ch"], s0["Q_per_ch"])
    m.profile(False)
    _, s2 = m.predict(t, r, dim_order=meta["dim_order"], frames_per_second=meta["fps"])
    assert "kernel_ms" not in s2


@pytest.mark.parametrize("W,H,F,disp,fuse_mode", [(683, 389, 3, "standard_fhd", 0), (1366, 768, 2, "standard_4k", 0), (250, 131, 1, "standard_fhd", 0),
                                                    # the FEAT instantiations of the fused band kernels (k_band4s_feat on the strips away from the border,
                                                    # k_band4f_feat on the border strips): several strips and segments, W % 4 == 2 with an odd height
                                                    (736, 416, 3, "standard_fhd", 1), (1446, 333, 2, "standard_hdr_pq", 1), (1200, 200, 3, "standard_4k", 1)])
def test_fused_features_on_ragged_multi_strip_frames(W, H, F, disp, fuse_mode):
    """The FEAT instantiation of k_band4 (column sums per piece of a cell row + k_feature_finish) where the small fixtures do not
    reach: widths that are not a multiple of 8 (RAGGED + FEAT), several strips (cells that straddle a strip seam), several row
    segments (cell rows cut into pieces), levels of odd size; against the oracle's feature pooling, which is pinned to the real
    reference's extract_features (tests/test_oracle_vs_golden.py)."""
    import colorvideovdp_amd as cv
    from oracle import cvvdp_oracle as orc
    rng = np.random.default_rng(W)
    y, x = np.mgrid[0:H, 0:W]
    ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (3.1 * x / W + f / 9.0) + c) * np.cos(2 * np.pi * 2.3 * y / H) for c in range(3)])
                    for f in range(F)], axis=1)[None]
    test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1)
    ref8, test8 = np.round(ref * 255).astype(np.uint8), np.round(test * 255).astype(np.uint8)
    fps = 0 if F == 1 else 30
    o = orc.Oracle(display_name=disp, features=True)
    _, ostats = o.predict(test8, ref8, dim_order="BCFHW", frames_per_second=fps)
    m = cv.cvvdp(display_name=disp, block_frames=2)
    m.fuse_mode = fuse_mode
    vs = cv.video_source_array(test8, ref8, fps, dim_order="BCFHW", display_photometry=m.display_photometry)
    feats, _ = m.extract_features(vs)
    assert (m.fused_levels >= 1) == (fuse_mode == 1)
    assert len(feats) == len(ostats["features"])
    for bb, (f, want) in enumerate(zip(feats, ostats["features"])):
        got = f.cpu().numpy()
        assert got.shape == want.shape, (bb, got.shape, want.shape)
        for q in (0, 2, 4):
            # (atol: the transient channel's coarse-band means are ~1e-4, sums of a few dozen rounding-level terms; 2.1e-6 observed)
            np.testing.assert_allclose(got[..., q], want[..., q], rtol=5e-4, atol=4e-6, err_msg=f"band {bb} mean {q}")
            scale = np.abs(want[..., q]) ** 2 + np.abs(want[..., q + 1])
            assert np.all(np.abs(got[..., q + 1] - want[..., q + 1]) <= 2e-3 * scale + 1e-7), f"band {bb} var {q + 1}"


# ---------------------------------------------------------------------------------------------------------------------
# k_band4f (band4f.hip): band kernels that compute the next pyramid level themselves.  Normal use picks them per clip for blocks that
# fill the GPU several times over (the bench clips: test_bench_clip_against_reference holds them to the real reference); here the
# test hook cvvdp.fuse_mode forces them on small frames, where the oracle is affordable and every border rule is close by.
def _fuse_clip(W, H, F, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W]
    ref = np.stack([np.stack([0.45 + 0.3 * np.sin(2 * np.pi * (3.1 * x / W + f / 9.0) + c) * np.cos(2 * np.pi * 2.3 * y / H) for c in range(3)])
                    for f in range(F)], axis=1)[None]
    test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1)
    return np.round(test * 255).astype(np.uint8), np.round(ref * 255).astype(np.uint8)


@pytest.mark.parametrize("W,H,F,fps,disp", [
    (256, 144, 5, 30, "standard_fhd"),        # one strip, levels 256x144 -> 128x72 -> 64x36 fused
    (736, 416, 3, 60, "standard_4k"),         # four strips (the last one narrow), several segments
    (512, 271, 4, 24, "standard_fhd"),        # odd height: the last coarse row's extra taps, the row-parity column edge (Q1)
    (1200, 90, 3, 50, "standard_4k"),         # W a multiple of the strip width, few rows: top and bottom mirrors in one segment
    (248, 600, 3, 60, "standard_hdr_pq"),     # a strip whose 256 columns end exactly at the image; tall
    (728, 120, 3, 60, "standard_fhd"),        # ... and not the first strip: lane 63 of strip 2 computes the LAST coarse column (round 5: a border strip)
    (152, 69, 12, 30, "standard_fhd"),        # segments of 14 rows: shorter than two blur radii, the last one 13 rows
    # W % 8 != 0 (even widths): the border strips' instantiation with a partial lane (W % 4 == 2) / the aligned one at W % 8 == 4
    (1366, 200, 3, 60, "standard_4k"),        # the laptop width: six strips, level 0 fused (683 columns at level 1: odd, reduce pass)
    (854, 97, 2, 30, "standard_fhd"),         # 480p width, odd height
    (254, 144, 4, 30, "standard_fhd"),        # two strips, the second 14 columns wide
    (486, 130, 3, 60, "standard_4k"),         # TWO strips reach the right border (x0 = 240: 488 > 486, x0 = 480: six columns)
    (38, 50, 3, 24, "standard_fhd"),          # one strip that is left and right border at once
    (250, 301, 2, 60, "standard_hdr_pq"),     # the first strip ends two columns short of the image; odd height
    (492, 100, 3, 30, "standard_fhd"),        # W % 8 == 4: level 0 on the aligned border kernel, level 1 (246) on the partial-lane one
    (36, 33, 2, 30, "standard_fhd"),          # W % 8 == 4, the smallest fusable frame there is
])
def test_fused_reduce_band_kernels(W, H, F, fps, disp):
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import _capi
    from oracle import cvvdp_oracle as orc
    t, r = _fuse_clip(W, H, F, W + H)
    ojod, ostats = orc.Oracle(display_name=disp).predict(t, r, dim_order="BCFHW", frames_per_second=fps)
    runs = {}
    for mode in (1, 2):
        m = cv.cvvdp(display_name=disp)
        m.fuse_mode = mode
        jod, stats = m.predict(t, r, dim_order="BCFHW", frames_per_second=fps)
        L = stats["Q_per_ch"].shape[-1]
        pyr = []
        hh, ww = H, W
        for l in range(L):
            pyr.append(m.debug_buffer(_capi.BUF_GPYR, l)[:8 * F * hh * ww].view(8, F, hh, ww).cpu().numpy().copy())
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
        runs[mode] = (float(jod), stats["Q_per_ch"], pyr)
    (j1, q1, p1), (j2, q2, p2) = runs[1], runs[2]
    # the pyramid: levels written by the band kernels against the reduce kernels' (same taps, another summation order)
    for l, (a, b) in enumerate(zip(p1, p2)):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-6 * max(1.0, float(np.abs(b).max())), err_msg=f"pyramid level {l}")
    assert not np.array_equal(p1[1], p2[1]) or W * H < 0      # (the two routes do round differently: the fused one did run)
    # (since round 6 the unfused route reduces levels <= 128 x 128 in the reference's operation order while the fused kernels -- forced onto
    # these small frames by the test hook -- keep the horizontal pass first: at a coarse band of a few dozen pixels the two now differ by
    # up to the generic tolerance, 1.1e-4 observed; both are held to the oracle below)
    np.testing.assert_allclose(q1, q2, rtol=2e-4, atol=2e-6)
    assert abs(j1 - float(ojod)) <= JOD_TOL and abs(j2 - float(ojod)) <= JOD_TOL
    np.testing.assert_allclose(q1, ostats["Q_per_ch"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("dt", ["u16", "f16", "f32"])
def test_fused_kernels_with_a_batch_and_other_sample_formats(dt):
    """The work items of the fused band kernels are (frame, batch entry) pairs; the sample format only concerns the temporal kernel in
    front of them.  A batch of two test clips against one broadcast reference (video_source.py:247-252), forced onto the fused kernels,
    against the oracle and against the unfused route."""
    import colorvideovdp_amd as cv
    from oracle import cvvdp_oracle as orc
    W, H, F = 502, 130, 4                                           # W % 4 == 2: three strips, the partial-lane border kernel
    t8, r8 = _fuse_clip(W, H, F, 77)
    rng = np.random.default_rng(78)
    t = np.concatenate([t8, np.clip(t8.astype(np.int32) + rng.integers(-6, 7, t8.shape), 0, 255).astype(np.uint8)], axis=0).astype(np.float64) / 255.0
    r = r8.astype(np.float64) / 255.0
    if dt == "u16":
        t, r = np.round(t * 65535).astype(np.uint16), np.round(r * 65535).astype(np.uint16)
    elif dt == "f16":
        t, r = torch.tensor(t.astype(np.float16)), torch.tensor(r.astype(np.float16))
    else:
        t, r = t.astype(np.float32), r.astype(np.float32)
    ojod, ostats = orc.Oracle(display_name="standard_hdr_pq" if dt == "u16" else "standard_4k").predict(t, r, dim_order="BCFHW", frames_per_second=50)
    qs = {}
    for mode in (1, 2):
        m = cv.cvvdp(display_name="standard_hdr_pq" if dt == "u16" else "standard_4k")
        m.fuse_mode = mode
        jod, stats = m.predict(t, r, dim_order="BCFHW", frames_per_second=50)
        assert m.fused_levels == (1 if mode == 1 else 0)
        assert np.all(np.abs(jod.cpu().numpy() - np.asarray(ojod)) <= JOD_TOL)
        np.testing.assert_allclose(stats["Q_per_ch"], ostats["Q_per_ch"], rtol=2e-4, atol=2e-6)
        qs[mode] = stats["Q_per_ch"]
    assert qs[1].shape[0] == 2
    np.testing.assert_allclose(qs[1], qs[2], rtol=2e-4, atol=2e-6)


def test_fused_route_is_a_property_of_the_clip():
    """Blocks and shards of a clip take the same kernels (the decision uses the nominal block): bit-identical for any blocking."""
    import colorvideovdp_amd as cv
    t, r = _fuse_clip(736, 416, 9, 5)
    qs = []
    for nb in (None, 4, 1):
        m = cv.cvvdp(display_name="standard_4k", block_frames=nb)
        m.fuse_mode = 1
        _, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=60)
        qs.append(st["Q_per_ch"])
    np.testing.assert_array_equal(qs[0], qs[1])
    np.testing.assert_array_equal(qs[0], qs[2])


# ---------------------------------------------------------------------------------------------------------------------
# k_band4s (band4s.hip): the fused band kernel with its work divided between front and back waves.  Same arithmetic, operation for
# operation, as the one-wave-per-channel k_band4f<4, 0> it replaces on the border-free strips: bits, not tolerances.
@pytest.mark.parametrize("W,H,F,fps,disp", [
    (1200, 144, 3, 60, "standard_4k"),        # five strips: three of them away from the border (front / back waves), two levels fused
    (736, 416, 5, 30, "standard_fhd"),        # four strips (the last one 16 columns), several row segments, odd frame count
    (1446, 333, 2, 60, "standard_hdr_pq"),    # W % 4 == 2 (partial-lane border kernel beside the split kernel), odd height
    (3840, 270, 2, 60, "standard_4k"),        # the bench clip's width: 14 of 16 strips on the split kernel
    (728, 96, 2, 30, "standard_fhd"),         # strip 2 ends exactly at the image: a border strip in both layouts
])
def test_split_band_kernel_matches_the_one_wave_layout(W, H, F, fps, disp):
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import _capi
    t, r = _fuse_clip(W, H, F, W - H)
    runs = {}
    for layout in (0, 1):
        m = cv.cvvdp(display_name=disp)
        m.fuse_mode, m.band_layout = 1, layout
        jod, stats = m.predict(t, r, dim_order="BCFHW", frames_per_second=fps)
        pyr = []
        hh, ww = H, W
        for l in range(min(3, stats["Q_per_ch"].shape[-1])):
            pyr.append(m.debug_buffer(_capi.BUF_GPYR, l)[:8 * F * hh * ww].view(8, F, hh, ww).cpu().numpy().copy())
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
        runs[layout] = (float(jod), stats["Q_per_ch"], pyr, m.fused_levels)
    assert runs[0][3] == runs[1][3] >= 1
    for l, (a, b) in enumerate(zip(runs[0][2], runs[1][2])):
        np.testing.assert_array_equal(a, b, err_msg=f"pyramid level {l}")
    # the per-frame sums: the same chain of operations in two separately compiled kernels -- the compiler contracts a multiply-add here
    # and not there, so single values may differ in the last bit (observed: one of 64 by one ulp); a clip is scored by ONE layout
    np.testing.assert_allclose(runs[0][1], runs[1][1], rtol=3e-7, atol=0)
    assert abs(runs[0][0] - runs[1][0]) < 2e-6


# ---------------------------------------------------------------------------------------------------------------------
# Heat-map clips resident in HBM (cvvdp_clip.defer_bands, cvvdp_score_frames): a long temporal block, the band / heat-map stage in
# pieces.  Neither length may change a bit of Q_per_ch or of the heat maps.
@pytest.mark.parametrize("mode,fuse_mode", [("supra-threshold", 0), ("raw", 0), ("threshold", 1)])     # fuse_mode 1: the pieces on the fused band kernels
def test_heatmap_clips_in_hbm_are_scored_in_pieces_of_a_long_temporal_block(mode, fuse_mode):
    import colorvideovdp_amd as cv
    t, r = _fuse_clip(256, 144, 41, 5)
    t, r = torch.as_tensor(t).cuda(), torch.as_tensor(r).cuda()
    runs = []
    #              block, piece -> (temporal block, piece; 0 = blocks scored whole)
    for block, piece, want in ((16, None, (16, 0)), (None, None, (41, 16)), (41, 7, (41, 7)), (23, 5, (23, 5)), (41, 41, (41, 0)), (30, 16, (30, 16))):
        m = cv.cvvdp(display_name="standard_fhd", heatmap=mode, block_frames=block)
        m.score_frames, m.fuse_mode = piece, fuse_mode
        jod, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=30)
        assert (m.last_block_frames, m.last_score_frames) == want and (m.fused_levels >= 1) == (fuse_mode == 1)
        runs.append((float(jod), st["Q_per_ch"], st["heatmap"].clone()))
    for jod, q, hm in runs[1:]:
        assert jod == runs[0][0]
        np.testing.assert_array_equal(q, runs[0][1])
        assert torch.equal(hm, runs[0][2])
    # the sink sees the pieces, in order
    order = []
    got = torch.zeros_like(runs[0][2])

    def sink(first, frames):
        order.append((first, frames.shape[2]))
        got[:, :, first:first + frames.shape[2]] = frames

    m = cv.cvvdp(display_name="standard_fhd", heatmap=mode, block_frames=23)
    m.score_frames, m.fuse_mode = 9, fuse_mode
    vs = cv.video_source_array(t, r, 30, dim_order="BCFHW", display_photometry=m.display_photometry)
    _, st = m.predict_video_source(vs, heatmap_sink=sink)
    # (a host sink: the first temporal block is one piece long, so that the D2H stream starts early; then blocks of 23 in pieces of 9)
    assert order == [(0, 9), (9, 9), (18, 9), (27, 5), (32, 9)]
    assert torch.equal(got, runs[0][2])
    np.testing.assert_array_equal(st["Q_per_ch"], runs[0][1])


def test_score_frames_argument_checks():
    import ctypes
    import colorvideovdp_amd as cv
    from colorvideovdp_amd import _capi
    t, r = _fuse_clip(64, 48, 12, 3)
    t, r = torch.as_tensor(t).cuda(), torch.as_tensor(r).cuda()
    m = cv.cvvdp(display_name="standard_fhd", heatmap="threshold", block_frames=12)
    m.score_frames = 4
    m.predict(t, r, dim_order="BCFHW", frames_per_second=30)
    lib, s = _capi.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.cvvdp_score_frames(m._handle, 0, 5, s) != 0              # more than score_frames
    assert lib.cvvdp_score_frames(m._handle, 10, 4, s) != 0             # beyond the frames the last block left
    assert lib.cvvdp_score_frames(m._handle, 8, 4, s) == 0
    m2 = cv.cvvdp(display_name="standard_fhd")
    m2.predict(t, r, dim_order="BCFHW", frames_per_second=30)
    assert lib.cvvdp_score_frames(m2._handle, 0, 1, s) != 0             # not configured with defer_bands


def test_device_heatmap_sink_gets_the_same_frames_without_pcie():
    """A sink with wants_device is handed the device tensor of every piece (fp16 planes or the writers' uint8 frames)."""
    import colorvideovdp_amd as cv
    t, r = _fuse_clip(256, 144, 20, 6)
    t, r = torch.as_tensor(t).cuda(), torch.as_tensor(r).cuda()
    m = cv.cvvdp(display_name="standard_fhd", heatmap="threshold", block_frames=20)
    m.score_frames = 8
    _, full = m.predict(t, r, dim_order="BCFHW", frames_per_second=30)
    vs = cv.video_source_array(t, r, 30, dim_order="BCFHW", display_photometry=m.display_photometry)
    for u8 in (False, True):
        seen = []

        class Sink:
            wants_device, wants_uint8 = True, u8

            def __call__(self, first, frames):
                assert frames.is_cuda
                seen.append((first, frames.clone()))

        _, st = m.predict_video_source(vs, heatmap_sink=Sink())
        assert [f for f, _ in seen] == [0, 8, 16]
        if u8:
            got = torch.cat([x for _, x in seen], 0).cpu()                                     # [n, H, W, 3] uint8
            from colorvideovdp_amd.heatmap_writers import heatmap_to_uint8
            np.testing.assert_array_equal(got.numpy(), heatmap_to_uint8(full["heatmap"]))
        else:
            assert torch.equal(torch.cat([x for _, x in seen], 2).cpu(), full["heatmap"])
        np.testing.assert_array_equal(st["Q_per_ch"], full["Q_per_ch"])


# ---------------------------------------------------------------------------------------------------------------------
# Heat-map clips on the fused band kernels (k_band4s_heat / k_band4f_heat): the band kernel computes the next level AND the level's
# heat-map band.  Against the unfused route: same heat map to fp16 rounding; between the two wave layouts: the same bits.
@pytest.mark.parametrize("W,H,F,fps,disp,mode", [
    (1200, 144, 3, 60, "standard_4k", "supra-threshold"),      # five strips, three on the split kernel
    (736, 416, 4, 30, "standard_fhd", "threshold"),            # several row segments
    (1446, 333, 2, 60, "standard_hdr_pq", "raw"),              # W % 4 == 2: partial-lane border kernel, odd height
    # frames on which EVERY strip is a border strip -- the context plane's range (tone curve, clamp) then comes only from the
    # edge-stream kernel's atomics, which must be ordered after the initialisation of the range words on the main stream
    (472, 240, 3, 60, "standard_fhd", "threshold"),            # two strips, both at the border
    (232, 176, 2, 60, "standard_fhd", "supra-threshold"),      # one strip with both borders
    (488, 150, 2, 60, "standard_fhd", "raw"),                  # W = 240 + 248: strip 1 ends exactly at the image (its lane 63 = the last coarse column)
])
def test_fused_band_kernels_write_the_heat_map_bands(W, H, F, fps, disp, mode):
    import colorvideovdp_amd as cv
    t, r = _fuse_clip(W, H, F, W + 3 * H)
    t, r = torch.as_tensor(t).cuda(), torch.as_tensor(r).cuda()
    runs = {}
    for name, fuse_mode, layout in (("unfused", 2, 0), ("split", 1, 0), ("one_wave", 1, 1)):
        m = cv.cvvdp(display_name=disp, heatmap=mode)
        m.fuse_mode, m.band_layout = fuse_mode, layout
        jod, st = m.predict(t, r, dim_order="BCFHW", frames_per_second=fps)
        assert (m.fused_levels >= 1) == (fuse_mode == 1)
        runs[name] = (float(jod), st["Q_per_ch"], st["heatmap"].clone())
    assert abs(runs["split"][0] - runs["one_wave"][0]) < 2e-6
    # (two separately compiled kernel families -- since round 5 the border strips too: k_band4s' EDGE body against k_band4f<4, 1> -- whose
    # fp32 partial sums differ in the last bits: observed 6.7e-7)
    np.testing.assert_allclose(runs["split"][1], runs["one_wave"][1], rtol=1.5e-6, atol=0)
    dl = (runs["split"][2].float() - runs["one_wave"][2].float()).abs()
    assert float(dl.max()) <= 1e-3 and float((dl > 0).float().mean()) < 1e-3      # (fp16 codes; the last-bit caveat of the test above)
    assert abs(runs["split"][0] - runs["unfused"][0]) < 1e-4
    np.testing.assert_allclose(runs["split"][1], runs["unfused"][1], rtol=2e-4, atol=2e-6)       # (test_fused_reduce_band_kernels' bound)
    d = (runs["split"][2].float() - runs["unfused"][2].float()).abs()
    # the two routes round the pyramid differently in the last bit; the map is fp16: a handful of pixels land on a neighbouring code
    assert float(d.max()) <= 2e-3 and float((d > 0).float().mean()) < 0.02, (float(d.max()), float((d > 0).float().mean()))
