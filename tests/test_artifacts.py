"""Experiment artefact writer against files written by the reference's own
harness for the same canned results (tests/golden/g7_artifacts.json): byte-identical."""

import json
import os

import pytest


def load():
    with open(os.path.join(os.path.dirname(__file__), "golden", "g7_artifacts.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("which", [0, 1])
def test_artifacts_byte_identical(tmp_path, which):
    from bimodalattack_amd.artifacts import ExperimentWriter
    from bimodalattack_amd.config import BimodalAttackResult
    g = load()
    exp = g["experiments"][which]
    base = tmp_path / "experiments"
    for _ in range(which):                                   # exp numbering continues from what exists
        os.makedirs(base / f"exp{_ + 1}")
    w = ExperimentWriter(exp["name"], g["config_kwargs"], [tuple(p) for p in exp["pairs"]], g["seed"], str(base))
    assert os.path.basename(w.folder) == exp["folder"]
    for i, c in enumerate(exp["canned"], start=1):
        w.images_folder(i)
        w.add(None if "raise" in c else BimodalAttackResult(**c))
    folder = w.close()
    for fn, want in exp["files"].items():
        got = open(os.path.join(folder, fn), newline="").read()
        assert got == want, fn
    assert sorted(d for d in os.listdir(folder) if os.path.isdir(os.path.join(folder, d))) == exp["dirs"]
    extra = set(os.listdir(folder)) - set(exp["files"]) - set(exp["dirs"]) - {"losses_aggregated.png"}
    assert not extra


def test_folder_numbering(tmp_path):
    from bimodalattack_amd.artifacts import next_experiment_folder
    base = str(tmp_path / "e")
    assert next_experiment_folder(base).endswith("exp1")
    os.makedirs(os.path.join(base, "exp7"))
    os.makedirs(os.path.join(base, "expX"))
    assert next_experiment_folder(base).endswith("exp8")
