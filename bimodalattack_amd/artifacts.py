"""Experiment artefacts: the on-disk contract of the reference's harness,
so its evaluation / demo tooling can consume this engine's runs unchanged.

Reference: experiments.py:54-285 (``run_experiment``) and utils/experiments_utils.py:26-71
(folder helpers, ``write_parameters_csv``).  One experiment folder ``<base>/expN/`` holds
``prompts.csv``, ``losses.csv``, ``details.csv``, ``times.csv``, ``parameters.csv``,
``best_strings.txt``, ``summary.csv`` and one ``images_<run>/`` folder per prompt (the PNGs
``run()`` writes every step).  Files are byte-identical to the reference's for the same
results (tests/golden/g7_artifacts.json); the loss plot is written only when matplotlib is
importable.
"""

from __future__ import annotations

import csv
import os
import time
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .config import BimodalAttackConfig, BimodalAttackResult

TIME_COLUMNS = ["Gradient Time", "Sampling Time", "PGD Time", "Loss Time", "Total Time"]


def next_experiment_folder(base: str = "experiments") -> str:
    """<base>/exp<max+1>, created (utils/experiments_utils.py:26-42)."""
    os.makedirs(base, exist_ok=True)
    top = 0
    for d in os.listdir(base):
        if d.startswith("exp") and os.path.isdir(os.path.join(base, d)):
            try:
                top = max(top, int(d[3:]))
            except ValueError:
                pass
    path = os.path.join(base, f"exp{top + 1}")
    os.makedirs(path, exist_ok=True)
    return path


def images_folder(exp_folder: str, run_index: int) -> str:
    p = os.path.join(exp_folder, f"images_{run_index}")
    os.makedirs(p, exist_ok=True)
    return p


def failed_result() -> BimodalAttackResult:
    """What the harness records for a prompt whose attack raised (experiments.py:116-137)."""
    return BimodalAttackResult(best_loss=float("nan"), best_string="", losses=[], strings=[], adversarial_suffixes=[],
                               model_outputs=[], gradient_times=[], sampling_times=[], pgd_times=[], loss_times=[],
                               total_times=[])


def _write_csv(path: str, header: Sequence, rows: Iterable[Sequence]) -> None:
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(header)
        w.writerows(rows)


def _cell(seq: Sequence, i: int):
    return seq[i] if i < len(seq) else ""


class ExperimentWriter:
    def __init__(self, name: str, config_kwargs: Dict, pairs: Sequence[Tuple[str, str]], seed: int = 1,
                 base: str = "experiments", folder: Optional[str] = None):
        self.name, self.config_kwargs, self.pairs, self.seed = name, dict(config_kwargs), list(pairs), seed
        self.folder = folder or next_experiment_folder(base)
        os.makedirs(self.folder, exist_ok=True)
        self.results: List[BimodalAttackResult] = []
        with open(os.path.join(self.folder, "prompts.csv"), "w", newline="", encoding="utf-8") as f:
            w = csv.writer(f)
            w.writerow(["Run", "goal", "target"])
            for i, (g, t) in enumerate(self.pairs, start=1):
                w.writerow([i, g, t])

    def images_folder(self, run_index: int) -> str:
        return images_folder(self.folder, run_index)

    def add(self, result: Optional[BimodalAttackResult]) -> None:
        """One finished prompt; ``None`` records a failed one."""
        self.results.append(result if result is not None else failed_result())

    def close(self) -> str:
        res, folder = self.results, self.folder
        n = len(res)
        losses = [r.losses for r in res]
        best = [r.best_loss for r in res]
        times = [[r.gradient_times, r.sampling_times, r.pgd_times, r.loss_times, r.total_times] for r in res]

        rows = max((len(l) for l in losses), default=0)
        _write_csv(os.path.join(folder, "losses.csv"), ["Iteration"] + [f"Run {i + 1}" for i in range(n)],
                   [[i] + [_cell(l, i) for l in losses] for i in range(rows)])

        rows = max((len(r.adversarial_suffixes) for r in res), default=0)
        header = ["Iteration"]
        for i in range(n):
            header += [f"Run {i + 1} Suffix", f"Run {i + 1} Output"]
        _write_csv(os.path.join(folder, "details.csv"), header,
                   [[i] + [c for r in res for c in (_cell(r.adversarial_suffixes, i), _cell(r.model_outputs, i))]
                    for i in range(rows)])

        rows = max((len(t[4]) for t in times), default=0)          # as many rows as the longest total_times
        header = ["Iteration"] + [f"Run {i + 1} {c}" for i in range(n) for c in TIME_COLUMNS]
        _write_csv(os.path.join(folder, "times.csv"), header,
                   [[i] + [_cell(col, i) for t in times for col in t] for i in range(rows)])

        with open(os.path.join(folder, "parameters.csv"), "w", newline="") as f:   # experiments_utils.py:51-71
            w = csv.writer(f)
            w.writerow(["Parameter", "Value"])
            w.writerow(["name", self.name])
            for k, v in self.config_kwargs.items():
                if k in ("alpha", "eps"):
                    w.writerow([k, self.config_kwargs.get(f"{k}_str", v)])
                elif not k.endswith("_str"):
                    w.writerow([k, v])
            w.writerow(["seed", self.seed])
            w.writerow(["num_prompts", len(self.pairs)])

        with open(os.path.join(folder, "best_strings.txt"), "w") as f:
            for i, r in enumerate(res, start=1):
                f.write(f"Run {i}: {r.best_string}\n")

        summary = [["Average Best Loss", np.mean(best) if best else float("nan")],
                   ["Std Best Loss", np.std(best) if best else float("nan")]]
        for k, label in enumerate(["Gradient", "Sampling", "PGD", "Loss", "Total"]):
            means = [np.mean(t[k]) if t[k] else float("nan") for t in times]
            summary += [[f"Average {label} Time", np.mean(means)], [f"Std {label} Time", np.std(means)]]
        _write_csv(os.path.join(folder, "summary.csv"), ["Metric", "Value"], summary)

        self._plot(losses)
        return folder

    def _plot(self, losses) -> None:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            return
        plt.figure(figsize=(10, 6), dpi=200)
        for i, l in enumerate(losses, start=1):
            plt.plot(l, linestyle="-", linewidth=1, label=f"Run {i}")
        plt.xlabel("Iteration")
        plt.ylabel("Loss")
        plt.title(self.name)
        text = "\n".join(f"{k}: {v}" for k, v in self.config_kwargs.items() if not k.endswith("_str"))
        plt.gca().text(0.98, 0.98, text, transform=plt.gca().transAxes, fontsize=8, va="top", ha="right",
                       bbox=dict(boxstyle="round", facecolor="white", alpha=0.5))
        plt.savefig(os.path.join(self.folder, "losses_aggregated.png"), bbox_inches="tight")
        plt.close()


def run_experiment(name: str, config_kwargs: Dict, pairs: Sequence[Tuple[str, str]], model, tokenizer, processor,
                   image=None, normalize=None, seed: int = 1, base: str = "experiments", **engine_options) -> str:
    """The reference harness's loop (experiments.py:54-152) on this engine: one attack per
    (goal, target) pair, failures recorded as NaN rows, artefacts written at the end."""
    import logging

    from .attack import run

    w = ExperimentWriter(name, config_kwargs, pairs, seed, base)
    for idx, (goal, target) in enumerate(pairs, start=1):
        cfg = BimodalAttackConfig(**{k: v for k, v in config_kwargs.items() if not k.endswith("_str") and k != "model"},
                                  seed=seed, verbosity="DEBUG", experiment_folder=w.folder,
                                  images_folder=w.images_folder(idx))
        try:
            t0 = time.time()
            result = run(model, tokenizer, processor, [{"role": "user", "content": goal}], goal, target, image, cfg,
                         normalize=normalize, **engine_options)
            logging.info(f"Run {idx} (Seed={seed}) -> Loss={result.best_loss:.4f}, Time={time.time() - t0:.2f}s")
        except Exception as e:
            logging.error(f"Error during attack for prompt {idx}/{len(pairs)}: {goal} -> {target}")
            logging.error(f"Exception: {e}")
            result = None
        w.add(result)
    return w.close()
