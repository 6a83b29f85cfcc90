"""CPU: the shape-keyed schedule switches of spvipes_amd.ops (no GPU, no library): which steps count as "small" (one stream + pair grids),
from where the label pairing forks beside the encoder tails, and where the mixture-weight GEMMs are held back.  The thresholds are measured
ones (docs/lab_notes.md E); this pins them so that a change shows up in review."""
import importlib


def _ops(monkeypatch, **env):
    for k in ("SPV_SERIAL_STREAMS", "SPV_DEC_PAIR", "SPV_LABEL_PRE", "SPV_WM_LATE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    import spvipes_amd.ops as ops
    return importlib.reload(ops)


def test_small_steps_run_on_one_stream_with_pair_grids(monkeypatch):
    ops = _ops(monkeypatch)
    for B, G, n, small in ((128, 2000, 2, True), (1024, 2000, 2, True), (128, 10_000, 2, True), (400, 10_000, 2, True), (512, 10_000, 2, False),
                           (4096, 10_000, 2, False), (128, 2000, 3, False), (128, 2000, 1, False)):
        ops.set_step_shape(B, G, n)
        assert ops.serial_streams() == small and ops.dec_pair_for(B, n) == small, (B, G, n)
    ops.set_step_shape(4096, 10_000, 2)


def test_environment_overrides(monkeypatch):
    ops = _ops(monkeypatch, SPV_SERIAL_STREAMS="1")
    ops.set_step_shape(4096, 10_000, 2)
    assert ops.serial_streams() and not ops.dec_pair_for(4096, 2)
    ops = _ops(monkeypatch, SPV_SERIAL_STREAMS="0")
    ops.set_step_shape(128, 2000, 2)
    assert not ops.serial_streams() and ops.dec_pair_for(128, 2)
    ops = _ops(monkeypatch, SPV_DEC_PAIR="0")
    ops.set_step_shape(128, 2000, 2)
    assert not ops.dec_pair_for(128, 2) and not ops.serial_streams()   # (one stream without the pair grids measured slower)
    ops = _ops(monkeypatch, SPV_DEC_PAIR="1")
    ops.set_step_shape(4096, 10_000, 2)
    assert ops.dec_pair_for(4096, 2) and not ops.serial_streams()
    _ops(monkeypatch)


def test_label_pairing_and_late_mixture_gemms_follow_the_shape(monkeypatch):
    ops = _ops(monkeypatch)
    assert ops.label_pre_for(128) == 0 and ops.label_pre_for(1023) == 0 and ops.label_pre_for(1024) == 2 and ops.label_pre_for(4096) == 2
    assert not ops.wm_late_for(10_000, False) and ops.wm_late_for(20_000, False) and not ops.wm_late_for(30_000, True)
    ops = _ops(monkeypatch, SPV_LABEL_PRE="1", SPV_WM_LATE="1")
    assert ops.label_pre_for(128) == 1 and ops.wm_late_for(100, True)
    _ops(monkeypatch)
