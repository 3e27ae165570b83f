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
    assert torch.equal(m.mask_c.cpu(), state["params.mask_c"]) and torch.equal(m.mask_q.cpu(), state["params.mask_q"])
    assert torch.equal(m.xcm_weights.cpu(), state["params.xcm_weights"]) and m.xcm_weights.dtype == torch.float32
    assert not hasattr(m, "head") and not hasattr(m, "epoch_marker")  # entries without the 'params.' prefix are skipped
    np.testing.assert_array_equal(m.ch_w.cpu().numpy(), np.float32([1.0, 1.1, 1.1, 0.9]))
    np.testing.assert_array_equal(m.baseband_weight.cpu().numpy(), np.float32([0.01, 1.5, 4.0, 20.0]))
    assert torch.is_tensor(m.jod_a) and m.jod_a.dtype == torch.float32       # every numeric parameter reads as a tensor, like the reference's


def _ckpt(path, **entries):
    torch.save({"state_dict": {"params." + k: torch.as_tensor(v, dtype=torch.float32) for k, v in entries.items()}}, path)
    return str(path)


def test_checkpoint_scalars_keep_their_fraction_and_bad_checkpoints_change_nothing(tmp_path):
    """(1) a scalar whose JSON literal is an integer (beta_tch: 4) takes a fractional checkpoint value as it is -- the core's
    parameters are all fp32; (2) a checkpoint that cannot be applied (wrong list length, unsupported beta, fractional pu_dilate) raises and
    leaves parameters, tensor views, derived weights and the core's handle as they were."""
    m = cv.cvvdp(display_name="standard_fhd")
    m.update_from_checkpoint(_ckpt(tmp_path / "a.ckpt", beta_tch=3.7, beta_t=2.25, ch_chrom_w=1.0))
    assert m.parameters["beta_tch"] == pytest.approx(3.7) and isinstance(m.parameters["beta_tch"], float)
    assert m.parameters["beta_t"] == 2.25 and float(m._params.beta_tch) == pytest.approx(3.7)
    before, v0, handle0 = dict(m.parameters), m._cfg_version, m._handle.value
    for bad in (dict(mask_q=[1.0, 2.0, 3.0]), dict(beta=3.0), dict(pu_dilate=2.5), dict(xcm_weights=[0.0] * 15), dict(mask_p=float("nan"))):
        with pytest.raises((RuntimeError, ValueError)):
            m.update_from_checkpoint(_ckpt(tmp_path / "bad.ckpt", mask_c=-0.5, **bad))
        assert m.parameters == before                                     # mask_c of the bad checkpoint did not get in either
        assert float(m.mask_c) == pytest.approx(before["mask_c"]) and float(m._params.mask_c10) == pytest.approx(10 ** before["mask_c"], rel=1e-6)
        assert m._handle.value                                            # a live handle (re-made from the old parameters if it had to be)
    np.testing.assert_array_equal(m.ch_w.cpu().numpy(), np.float32([1.0, 1.0, 1.0, before["ch_trans_w"]]))


def test_assigning_a_parameter_attribute_reconfigures_the_core(tmp_path):
    """`metric.mask_c = tensor` is the reference's way to change a parameter; here it lands in self.parameters and the
    handle is re-made (or it raises and nothing changes) -- it never creates a shadowing attribute the kernels do not see."""
    m = cv.cvvdp(display_name="standard_fhd")
    v0 = m._cfg_version
    m.mask_c = torch.tensor(-0.5)
    assert m.parameters["mask_c"] == -0.5 and "mask_c" not in m.__dict__ and m._cfg_version > v0
    assert float(m._params.mask_c10) == pytest.approx(10 ** -0.5, rel=1e-6)
    m.baseband_weight = [0.5, 1.0, 2.0, 3.0]
    np.testing.assert_array_equal(m._baseband_weight, np.float32([0.5, 1.0, 2.0, 3.0]))
    with pytest.raises(ValueError):
        m.mask_q = torch.ones(3)
    with pytest.raises(ValueError):
        m.pu_dilate = 1.5
    with pytest.raises(RuntimeError):
        m.pu_dilate = 2
    assert m.parameters["pu_dilate"] == 3 and len(m.parameters["mask_q"]) == 4
    out = tmp_path / "p.json"
    m.save_to_config(str(out), "x")
    assert json.loads(out.read_text())["mask_c"] == -0.5                  # save_to_config writes what the kernels run with


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
