"""Pre-tuned GEMM solution table for the nn.Linear shapes of the dual-branch step.

The Linear layers of the path run on hipBLASLt through torch.  hipBLASLt's default
heuristic is not the best solution for every shape of this model, so the solutions were selected once on an
MI355X with PyTorch TunableOp (``PYTORCH_TUNABLEOP_TUNING=1``, see ``tools/tune_gemms.sh``) and are shipped as
data; ``enable()`` loads them with tuning switched OFF, so nothing is tuned at run time.  The table is validated
by torch against the ROCm / hipBLASLt / GPU-arch versions recorded in its header and ignored if they differ.
"""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950_cfg2_bf16.csv")


def enable(table=TABLE):
    """Use the shipped GEMM solution table (no run-time tuning).  Returns True when it was loaded."""
    if not torch.cuda.is_available() or not os.path.exists(table):
        return False
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.record_untuned_enable(False)
    try:
        return bool(tunable.read_file(table))
    except Exception:            # version mismatch etc.: fall back to hipBLASLt's heuristic
        tunable.enable(False)
        return False
