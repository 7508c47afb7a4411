import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return {k: torch.from_numpy(np.asarray(v).astype(np.float32) if v.dtype == np.float16 else np.asarray(v))
            for k, v in np.load(os.path.join(GOLDEN, name)).items()}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def op_inputs():
    """Inputs of the per-op goldens (must match oracle/tools/gen_golden.py:op_inputs)."""
    g = torch.Generator().manual_seed(11)
    return dict(
        x=torch.randn(2, 64, 4, 8, 16, generator=g), emb=torch.randn(2, 256, generator=g),
        ctx=torch.randn(2, 141, 1024, generator=g), x3=torch.randn(2, 256, 4, 4, 8, generator=g),
        lat9=torch.randn(2, 9, 4, 8, 16, generator=g), feat=torch.randn(2, 16, 4096, 256, generator=g),
        px=torch.randn(40, 64, 2, 8, 8, generator=g), ex=torch.randn(2, 64, 2, 16, 32, generator=g))


def record(name, **vals):
    """Observed errors of the GPU parity tests: printed (pytest -s) and collected in gpurun_out/parity_observed.json, from
    where they go into profiles/ and DESIGN.md -- the stated tolerances come from measurement."""
    import json
    print("PARITY", name, {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in vals.items()}, flush=True)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_observed.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.isfile(path) else {}
        data[name] = vals
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass
