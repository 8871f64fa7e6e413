"""Host logic of the attack loop that needs no device: which segments make up a
candidate sequence in which mode (reference bimodal_attack.py:1150-1215), the
dynamic search-width schedule (:919-928) and where the shared prefix ends.

Table-driven: a layout is looked up, not branched to.
"""

from __future__ import annotations

from typing import List, Sequence, Tuple

# Sequence layouts WITH the image, keyed by "is the model gemma3".  Gemma-3 puts
# the suffix in front of the image, everything else behind it (:1153-1162).
_WITH_IMAGE = {
    False: ("before_img", "image", "before_suffix", "optim", "after"),
    True: ("before_img", "optim", "before_suffix", "image", "after"),
}
# Layouts of the PGD prompt scored WITHOUT the image (joint_eval=False, :1166-1171).
_NO_IMAGE = {
    False: ("before_img", "before_suffix", "optim", "after"),
    True: ("before_img", "optim", "before_suffix", "after"),
}
_TEXT_ONLY = ("before", "optim", "after")


def segment_order(mode: str, model_type: str, single: bool = False, no_joint_eval: bool = False,
                  no_target: bool = False) -> List[str]:
    """Names of the segments of one candidate sequence, in order.

    mode "pgd" / "gcg_pgd": prompt with image features; "gcg": text only, either the
    PGD prompt minus the image (``single``) or the GCG-only prompt (``no_joint_eval``
    with the target, ``no_target`` without).  Flag precedence follows the reference:
    ``single`` beats ``no_joint_eval`` beats ``no_target``.
    """
    gemma = model_type == "gemma3"
    if mode == "pgd":
        assert single, "PGD mode only supports single=True"
        return [*_WITH_IMAGE[gemma], "target"]
    if mode == "gcg_pgd":
        drop_target = no_target and not single
        return list(_WITH_IMAGE[gemma]) + ([] if drop_target else ["target"])
    if mode == "gcg":
        if single:
            return [*_NO_IMAGE[gemma], "target"]
        if no_joint_eval:
            return [*_TEXT_ONLY, "target"]
        if no_target:
            return list(_TEXT_ONLY)
        raise ValueError("Invalid flags for BimodalAttack mode")
    raise ValueError(f"Unknown mode '{mode}'")


def dynamic_width(step: int, search_width: int, num_steps: int, min_search_width: int, dynamic: bool) -> int:
    """Candidates to sample at `step`: linear decay to a floor when dynamic_search (:919-923)."""
    if not dynamic:
        return search_width
    return max(min_search_width, int(search_width * (1 - step / num_steps)))


def split_at_suffix(order: Sequence[str]) -> Tuple[List[str], List[str]]:
    """(segments in front of the suffix, suffix and everything behind it).  The first
    part is identical for every candidate of a step, so under causal attention its
    keys/values can be computed once and shared."""
    i = list(order).index("optim")
    return list(order[:i]), list(order[i:])
