"""CPU: the parts of spVIPES.save / spVIPES.load (docs/notebooks/Tutorial.ipynb:487,538; scvi-tools BaseModelClass) that need no GPU --
argument errors and the layout of the saved file.  The round trip itself is tests/test_gpu_train_and_model.py::test_save_and_load_round_trip."""
import inspect

import pytest

from spvipes_amd.model import spVIPES
from tests._duck import make_duck


def test_load_errors_come_before_any_device_work(tmp_path):
    with pytest.raises(ValueError, match="Failed to load model file"):
        spVIPES.load(str(tmp_path / "missing"), adata=make_duck())
    (tmp_path / "m").mkdir()
    (tmp_path / "m" / "model.pt").write_bytes(b"")
    with pytest.raises(ValueError, match="no adata was passed"):
        spVIPES.load(str(tmp_path / "m"))


def test_signatures_follow_scvi_base_model_class():
    s = inspect.signature(spVIPES.save)
    assert list(s.parameters)[:5] == ["self", "dir_path", "prefix", "overwrite", "save_anndata"]
    assert s.parameters["overwrite"].default is False and s.parameters["save_anndata"].default is False
    l = inspect.signature(spVIPES.load)
    assert list(l.parameters)[:5] == ["dir_path", "adata", "use_gpu", "prefix", "backup_url"] and isinstance(inspect.getattr_static(spVIPES, "load"), classmethod)
