"""What arithmetic and which solver a caller got is SAID: an unmodified reference-form config -- no
`precision` key, inline callables -- gets the fp32-emulating bf16x3 products and, by behavioural probing, the device solver; both
decisions are logged at WARNING level with the way back to the reference's behaviour, and an explicit choice is logged at INFO.
Host logic only: the engine (which needs the HIP device) is replaced by a stub, nothing is computed."""
import logging
import sys
from pathlib import Path

import pytest

sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from pytracking.utils.config import load_config  # noqa: E402
from woft_amd import flow_provider, tracker  # noqa: E402


class _StubEngine:
    def __init__(self, state_dict, **kw):
        self.kw = kw
        self.corr = kw.get("corr")


@pytest.fixture
def stubbed(monkeypatch):
    monkeypatch.setattr(flow_provider, "RaftEngine", _StubEngine)
    monkeypatch.setattr(tracker.YAOFTrackerSingleControl, "DEVICE", "cpu")
    monkeypatch.delenv("WOFT_PRECISION", raising=False)
    monkeypatch.delenv("WOFT_FUSED", raising=False)


def _records(caplog, name, needle):
    return [r for r in caplog.records if r.name == name and needle in r.getMessage()]


def test_reference_form_config_logs_precision_and_solver_at_warning_level(stubbed, caplog):
    conf = load_config(ROOT / "tests" / "configs" / "inline_wlsq.py")
    conf.flow_config.model = {}
    assert not conf.flow_config.precision                   # (a reference flow config has no such key)
    with caplog.at_level(logging.INFO):
        trk = conf.tracker_class(conf)
    (rec,) = _records(caplog, "woft_amd.flow_provider", "RAFT arithmetic")
    msg = rec.getMessage()
    assert rec.levelno == logging.WARNING
    assert "precision = 'bf16x3'" in msg and "built-in default" in msg and "precision = 'fp32'" in msg and "WOFT_PRECISION=fp32" in msg
    assert trk.flower.precision == "bf16x3" and trk.flower.precision_source.startswith("built-in default")
    (rec,) = _records(caplog, "woft_amd.tracker", "tracker solver")
    msg = rec.getMessage()
    assert rec.levelno == logging.WARNING
    assert "device back end" in msg and msg.count("probed") >= 3 and "device_solver = False" in msg
    assert trk.solver_decision.startswith("device back end")


def test_explicit_choices_are_logged_at_info_level(stubbed, caplog):
    conf = load_config(ROOT / "pytracking" / "configs" / "WOFT.py")          # the shipped config: precision key + tagged presets
    conf.flow_config.model = {}
    with caplog.at_level(logging.INFO):
        trk = conf.tracker_class(conf)
    (rec,) = _records(caplog, "woft_amd.flow_provider", "RAFT arithmetic")
    assert rec.levelno == logging.INFO and "flow config key 'precision'" in rec.getMessage()
    (rec,) = _records(caplog, "woft_amd.tracker", "tracker solver")
    assert rec.levelno == logging.INFO and "tagged" in rec.getMessage() and "probed" not in rec.getMessage()
    assert trk.solver_decision.startswith("device back end")


def test_exact_fp32_is_named_as_the_references_arithmetic(stubbed, caplog):
    conf = load_config(ROOT / "tests" / "configs" / "inline_wlsq.py")
    conf.flow_config.model = {}
    conf.flow_config.precision = "fp32"
    conf.device_solver = False
    with caplog.at_level(logging.INFO):
        trk = conf.tracker_class(conf)
    (rec,) = _records(caplog, "woft_amd.flow_provider", "RAFT arithmetic")
    assert rec.levelno == logging.INFO and "reference's arithmetic (exact IEEE fp32 products)" in rec.getMessage()
    assert trk.solver_decision.startswith("callable back end")
