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


def check_masks_64x128x32(tag, e2p, p2e, row_tol=1e-3, sum_rtol=2e-6):
    """One variant ("normal" / "oppo") of the cross-view bias matrices at the level-1 WarpAttn size of BASELINE cfg5 (equirect
    64 x 128, 20 views of 32 x 32: e2p [8192, 20480], p2e [20480, 8192], fp32 on any device) against
    tests/golden/masks_64x128x32.npz -- the REAL get_merged_masks (src/utils/utils.py:12-41; oracle/tools/gen_golden.py masks5):
    48 sampled rows of each matrix (fp16 in the fixture) and, over ALL entries, per-row / per-column sums of (mask + 1), the
    background share and the value range.  Returns the observed deviations."""
    g = gold("masks_64x128x32.npz")
    obs = {}
    for name, mat, rows in (("e2p", e2p, g["rows_e2p"]), ("p2e", p2e, g["rows_p2e"])):
        mat = mat.float()
        obs[f"{name}_rows_max_abs"] = float((mat[rows.to(mat.device)].cpu() - g[f"{name}_{tag}_rows"].float()).abs().max())
        d = mat.double() + 1.0
        rs, cs = g[f"{name}_{tag}_rowsum"], g[f"{name}_{tag}_colsum"]
        obs[f"{name}_rowsum_rel"] = float((d.sum(dim=1).cpu() - rs).abs().max() / rs.abs().max())
        obs[f"{name}_colsum_rel"] = float((d.sum(dim=0).cpu() - cs).abs().max() / cs.abs().max())
        obs[f"{name}_background_diff"] = abs(float((mat == -1).double().mean()) - float(g[f"{name}_{tag}_background"]))
        lo, hi = (float(v) for v in g[f"{name}_{tag}_minmax"])
        assert float(mat.min()) == lo and float(mat.max()) == hi, (name, tag, float(mat.min()), float(mat.max()))
        assert obs[f"{name}_rows_max_abs"] < row_tol, (name, tag, obs)
        assert obs[f"{name}_rowsum_rel"] < sum_rtol and obs[f"{name}_colsum_rel"] < sum_rtol, (name, tag, obs)
        assert obs[f"{name}_background_diff"] < 1e-6, (name, tag, obs)
    return obs
