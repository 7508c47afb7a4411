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


STATUS = {"applied": False, "reason": "enable() not called"}


def _table_validators(table):
    """The Validator lines of the CSV header: what the table was tuned against (ROCm / hipBLASLt / arch versions)."""
    out = []
    try:
        with open(table) as f:
            for line in f:
                if line.startswith("Validator,"):
                    out.append(line.strip()[len("Validator,"):])
    except OSError:
        pass
    return out


def enable(table=TABLE):
    """Use the shipped GEMM solution table (no run-time tuning).  Returns True when it was loaded; ``STATUS`` says why not
    otherwise -- and a rejected table is reported on stderr, never silently: the hipBLASLt share of the step is ~10 % slower
    on the default heuristic."""
    import sys
    global STATUS
    if not torch.cuda.is_available() or not os.path.exists(table):
        STATUS = {"applied": False, "reason": "no GPU" if not torch.cuda.is_available() else f"{table} missing"}
        return False
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.record_untuned_enable(False)
    try:
        ok = bool(tunable.read_file(table))
        reason = "loaded" if ok else "torch rejected the table (validator mismatch: ROCm / hipBLASLt / GPU architecture differ from the header)"
    except Exception as e:            # noqa: BLE001
        ok, reason = False, f"read_file raised {type(e).__name__}: {e}"
    if not ok:
        tunable.enable(False)
        have = []
        try:
            have = [",".join(map(str, v)) for v in tunable.get_validators()]
        except Exception:             # noqa: BLE001
            pass
        reason += f"; table tuned against {_table_validators(table)}, this runtime reports {have}"
        print(f"imagine360_amd.tuning: pre-tuned GEMM table NOT applied -- {reason}; hipBLASLt runs on its default heuristic "
              "(re-tune with tools/tune_gemms.sh)", file=sys.stderr)
    STATUS = {"applied": ok, "reason": reason}
    return ok
