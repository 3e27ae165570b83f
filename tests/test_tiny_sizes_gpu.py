"""Out-of-bounds pass over the kernels at the sizes where every border rule fires at once (tools/fuzz_tiny.py promoted):
frames of a few pixels -- narrower than a 16-byte access, than the 5-tap reduce, than the 13-tap blur, than one strip -- as images
and as 3-frame clips.  The HIP path must agree with the oracle wherever the oracle (= the reference's algorithm) produces a result,
and must refuse with an exception -- not crash, not return numbers -- where the oracle refuses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
JOD_TOL = 1e-3
SIZES = [(4, 4), (5, 7), (8, 8), (9, 16), (16, 9), (12, 33), (33, 12), (7, 64), (64, 7), (15, 15), (6, 6), (3, 9), (2, 2), (1, 8), (17, 31), (31, 17)]


@pytest.mark.parametrize("F,fps", [(1, 0), (3, 30)])
def test_tiny_frames_agree_with_the_oracle_or_are_refused(F, fps):
    import colorvideovdp_amd as cv
    from oracle import cvvdp_oracle as orc
    rng = np.random.default_rng(5)
    compared = 0
    for (H, W) in SIZES:
        ref = rng.random((1, 3, F, H, W)).astype(np.float32)
        test = np.clip(ref + 0.05 * rng.standard_normal(ref.shape), 0, 1).astype(np.float32)
        try:
            oj, os_ = orc.Oracle(display_name="standard_fhd").predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
        except Exception:
            with pytest.raises(Exception):
                cv.cvvdp(display_name="standard_fhd").predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
            continue
        j, s = cv.cvvdp(display_name="standard_fhd").predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
        assert s["Q_per_ch"].shape == os_["Q_per_ch"].shape, (W, H)
        assert np.isfinite(s["Q_per_ch"]).all(), (W, H)
        assert abs(float(j) - float(oj)) <= JOD_TOL, (W, H, float(j), float(oj))
        np.testing.assert_allclose(s["Q_per_ch"], os_["Q_per_ch"], rtol=2e-4, atol=2e-6, err_msg=f"{W}x{H}x{F}")
        compared += 1
    assert compared >= 10
