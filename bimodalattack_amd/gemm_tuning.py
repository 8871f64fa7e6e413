"""Library-GEMM selection for the HuggingFace model's matmuls.

The GEMMs of the candidate-scoring forward -- 82 % of a step -- stay in rocBLAS / hipBLASLt (north_star assigns
them to the library; the batch-1 gradient pass has hand-written products of its own since rounds 3-4: csrc/gemm_nt.hip
at <= 96 rows, csrc/gemm_mid.hip at 560-672 rows, routed in ops.linear_b1 and untouched by this module).
What this module controls is WHICH library kernel serves each shape.  PyTorch's TunableOp
can time every rocBLAS and hipBLASLt solution for a shape once and remember the winner;
``tools/tune_gemms.py`` does that offline for the attack's shapes (LLaVA-1.5-7B, sw=512,
1/2/4/8 shards) and the result ships as ``bimodalattack_amd/tuning/<arch>.csv``.  At run
time the file is loaded in LOOKUP-ONLY mode: no tuning, no timing, unknown shapes fall
through to the library's own heuristic.  The file carries validators (torch, HIP,
hipBLASLt, rocBLAS versions, GPU arch); on any mismatch PyTorch rejects it and nothing
changes.
"""

from __future__ import annotations

import logging
import os

import torch

logger = logging.getLogger("gcg")
_DONE = {"state": None}


def tuned_file(device) -> str:
    arch = torch.cuda.get_device_properties(device).gcnArchName.split(":")[0]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", f"{arch}.csv")


def enable(mode: str, device) -> bool:
    """Returns True when a tuned selection is active.  Idempotent; respects a user who has
    configured TunableOp through the environment (then this module does nothing)."""
    if _DONE["state"] is not None:
        return _DONE["state"]
    ok = False
    try:
        if mode == "off" or os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is not None:
            return ok
        path = tuned_file(device) if mode == "auto" else mode
        if not os.path.exists(path):
            return ok
        import torch.cuda.tunable as tun
        tun.enable(True)
        tun.tuning_enable(False)          # lookup only: never time anything at run time
        try:
            tun.record_untuned_enable(False)
        except Exception:
            pass
        ok = bool(tun.read_file(path))
        if not ok:
            tun.enable(False)
            logger.warning(f"GEMM selection file {path} rejected (library versions differ); using library heuristics")
    except Exception as e:  # never let a tuning nicety break the attack
        logger.warning(f"GEMM selection not enabled: {type(e).__name__}: {e}")
        ok = False
    finally:
        _DONE["state"] = ok
    return ok
