"""cvvdp.update_from_checkpoint / cvvdp.save_to_config (cvvdp_metric.py:231-243, 1129-1154) against the REAL
reference: tests/golden/checkpoint.npz (oracle/make_goldens_checkpoint.py) holds a synthetic calibration `state_dict`, the JSON the
reference's save_to_config wrote before / after loading it, and the reference's JOD / Q_per_ch on a committed video and image case with
the shipped and with the loaded parameters."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

import colorvideovdp_amd as cv


@pytest.fixture(scope="module")
def ck():
    return np.load(os.path.join(GOLDEN, "checkpoint.npz"), allow_pickle=False)


def _write_checkpoint(ck, path):
    state = {str(k): torch.from_numpy(np.array(ck["state/" + str(k)])) for k in ck["state_keys"]}
    torch.save({"state_dict": state, "epoch": 3}, path)
    return state


def _json_equal_but_date(a, b):
    a, b = json.loads(a), json.loads(b)
    a.pop("calibration_date"), b.pop("calibration_date")
    assert list(a) == list(b)                    # same keys in the same order: the layout of the shipped parameter file
    for k in a:
        assert type(a[k]) is type(b[k]), k
        if isinstance(a[k], (float, list)):
            np.testing.assert_array_equal(np.asarray(a[k], dtype=np.float64), np.asarray(b[k], dtype=np.float64), err_msg=k)
        else:
            assert a[k] == b[k], k


def test_save_to_config_and_checkpoint_parameters_match_the_reference(ck, tmp_path):
    m = cv.cvvdp(display_name="standard_fhd")
    out = tmp_path / "p.json"
    m.save_to_config(str(out), "as shipped")
    _json_equal_but_date(out.read_text(), str(ck["saved_before"]))
    with pytest.raises(AssertionError):
        m.save_to_config(str(tmp_path / "p.txt"), "x")
    with pytest.raises(AssertionError):
        m.update_from_checkpoint(str(tmp_path / "missing.ckpt"))
    state = _write_checkpoint(ck, str(tmp_path / "ck.ckpt"))
    v0 = m._cfg_version
    m.update_from_checkpoint(str(tmp_path / "ck.ckpt"))
    assert m._cfg_version > v0                                       # the core's handle was re-made: cached clip plans are stale
    m.save_to_config(str(out), "after the checkpoint")
    _json_equal_but_date(out.read_text(), str(ck["saved_after"]))
    assert json.loads(out.read_text())["__comment"] == "after the checkpoint"
    # the reference keeps its parameters as tensor attributes; here they read through to self.parameters
    assert torch.equal(m.mask_c, state["params.mask_c"]) and torch.equal(m.mask_q, state["params.mask_q"])
    assert torch.equal(m.xcm_weights, state["params.xcm_weights"]) and m.xcm_weights.dtype == torch.float32
    assert not hasattr(m, "head") and not hasattr(m, "epoch_marker")  # entries without the 'params.' prefix are skipped
    np.testing.assert_array_equal(m.ch_w, np.float32([1.0, 1.1, 1.1, 0.9]))
    np.testing.assert_array_equal(m.baseband_weight, np.float32([0.01, 1.5, 4.0, 20.0]))


@pytest.mark.gpu
def test_scores_after_update_from_checkpoint_match_the_reference(ck, tmp_path):
    _write_checkpoint(ck, str(tmp_path / "ck.ckpt"))
    m = cv.cvvdp(display_name="standard_fhd")
    for when in ("before", "after"):
        if when == "after":
            m.update_from_checkpoint(str(tmp_path / "ck.ckpt"))
        for tag in ("vid", "img"):
            g = np.load(os.path.join(GOLDEN, str(ck[f"{tag}_case"]) + ".npz"), allow_pickle=False)
            meta = eval(str(g["meta"]))
            m.set_display_model(meta["display"])
            m.temp_padding = meta["temp_padding"]
            jod, stats = m.predict(g["test"], g["ref"], dim_order=meta["dim_order"], frames_per_second=meta["fps"])
            assert abs(float(jod) - float(ck[f"{tag}_jod_{when}"])) <= 1e-3, (tag, when)     # north_star's tolerance; observed ~1e-5
            np.testing.assert_allclose(stats["Q_per_ch"], ck[f"{tag}_q_{when}"], rtol=2e-4, atol=2e-6, err_msg=f"{tag} {when}")
    assert abs(float(ck["vid_jod_after"]) - float(ck["vid_jod_before"])) > 0.1               # the checkpoint does move the score
