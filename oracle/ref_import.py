"""Import the reference's OWN classes from /root/reference (build container only; TEST INFRASTRUCTURE).

The reference scripts import ``dac``, ``soundfile``, ``torchaudio`` and ``matplotlib`` at module scope and create
``/home/student/...`` directories; none of that is needed for the classes on the hot path, so inert stand-ins are
placed in ``sys.modules`` and ``os.makedirs`` is patched while the module body runs.
Nothing here is used on the GPU box: ``tests/golden/make_golden.py`` calls it once to generate fixtures.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from pathlib import Path
from unittest import mock

REF_ROOT = Path("/root/reference")


def available() -> bool:
    return (REF_ROOT / "Training" / "compare_dacvsproposal_5.py").exists()


def _stub(name):
    m = mock.MagicMock(name=name)
    m.__name__ = name
    m.__spec__ = importlib.util.spec_from_loader(name, loader=None)
    return m


def load(rel_path: str, alias: str):
    """Execute one reference script as a module and return it (e.g. 'Training/compare_dacvsproposal_5.py')."""
    path = REF_ROOT / rel_path
    os.environ.setdefault("MPLBACKEND", "Agg")
    stubs = {}
    for name in ("dac", "soundfile", "torchaudio", "torchaudio.transforms", "torchaudio.functional",
                 "matplotlib", "matplotlib.pyplot", "h5py", "skimage", "skimage.metrics"):
        if name not in sys.modules:
            stubs[name] = _stub(name)
    import torch  # noqa: F401  (must be fully imported before any patching)
    spec = importlib.util.spec_from_file_location(alias, str(path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules.update(stubs)
    try:
        with mock.patch("os.makedirs"), mock.patch("torch.cuda.is_available", return_value=False):
            spec.loader.exec_module(mod)
    finally:
        for name in stubs:                       # remove only the stand-ins, keep everything really imported
            sys.modules.pop(name, None)
    return mod


def training():
    return load("Training/compare_dacvsproposal_5.py", "ref_train5")


def training3():
    return load("Training/compare_dacvsproposal_3.py", "ref_train3")


def evaluation():
    return load("Evaluation/dac_vcpwq_proposed6_latency.py", "ref_eval6")


def eval5():
    return load("Evaluation/compare_dacvsproposal_5_eval.py", "ref_eval5")


def eval35():
    return load("Evaluation/compare_dacvsproposal_3.5_eval.py", "ref_eval35")
