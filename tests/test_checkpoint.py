"""Checkpoint semantics of models/utils/checkpoint.py:11-120 on local files (CPU only: construction, state dicts and
file I/O do not touch the HIP path)."""
import collections
import logging

import pytest
import torch

import torch_detection_amd as T
from torch_detection_amd import checkpoint as C


def _model():
    return torch.nn.Sequential(collections.OrderedDict(a=torch.nn.Linear(3, 2), b=torch.nn.BatchNorm1d(2)))


def test_nonstrict_reports_and_strict_raises(capsys, caplog):
    m = _model()
    sd = collections.OrderedDict((k, torch.full_like(v, 2)) for k, v in m.state_dict().items())
    del sd["b.running_var"]
    sd["extra.weight"] = torch.zeros(1)
    C.load_state_dict(m, sd)                       # tolerated, printed
    out = capsys.readouterr().out
    assert "unexpected key in source state_dict: extra.weight" in out
    assert "missing keys in source state_dict: b.running_var" in out
    assert float(m.a.weight.detach()[0, 0]) == 2.0          # the matching keys were copied
    with caplog.at_level(logging.WARNING):
        C.load_state_dict(m, sd, logger=logging.getLogger("t"))
    assert "extra.weight" in caplog.text
    with pytest.raises(RuntimeError, match="unexpected key"):
        C.load_state_dict(m, sd, strict=True)


def test_shape_mismatch_raises():
    m = _model()
    sd = m.state_dict()
    sd["a.weight"] = torch.zeros(5, 5)
    with pytest.raises(RuntimeError, match="a.weight"):
        C.load_state_dict(m, sd)


def test_file_roundtrip_prefix_and_remote(tmp_path):
    m = T.ResNet(18)
    f = str(tmp_path / "sub" / "r18.pth")
    T.save_checkpoint(m, f, meta={"epoch": 3})
    ck = torch.load(f, weights_only=True)
    assert ck["meta"]["epoch"] == 3 and "time" in ck["meta"]
    assert all(v.is_contiguous() and v.device.type == "cpu" for v in ck["state_dict"].values())
    # DataParallel-style prefix is stripped; a bare OrderedDict is accepted
    pref = collections.OrderedDict(("module." + k, v) for k, v in ck["state_dict"].items())
    g = str(tmp_path / "pref.pth")
    torch.save(pref, g)
    m2 = T.ResNet(18)
    T.load_checkpoint(m2, g, strict=True)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(TypeError):
        T.save_checkpoint(m, f, meta=3)
    for url in ("modelzoo://resnet50", "https://example.org/x.pth"):
        with pytest.raises(IOError, match="remote"):
            T.load_checkpoint(m2, url)
    with pytest.raises(IOError):
        T.load_checkpoint(m2, str(tmp_path / "nope.pth"))
