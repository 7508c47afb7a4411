"""Which device kernels make up ONE eager cfg2 denoising step, ours and the libraries' (hipBLASLt GEMMs, torch
elementwise / copy kernels)?  bench.py's per-class table only covers the im360_* launches; this lists the rest.

    python tools/step_profile.py [--out gpurun_out/step_kernels.txt] [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagine360_amd import configs, kernels, synthetic          # noqa: E402
from imagine360_amd.scheduler import DDIMScheduler                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/step_kernels.txt")
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--width-div", type=int, default=1)
    args = ap.parse_args()
    dev, dt = torch.device("cuda", 0), torch.bfloat16
    torch.set_grad_enabled(False)
    kernels.lib()
    from imagine360_amd import tuning
    tuning.enable()
    w = dict(frames=16, pano_hw=(64, 128), pers_hw=(32, 32), pers_px=256)         # bench.py WORKLOADS["cfg2"]
    mv = configs.build_mv_model(args.width_div, device=dev, dtype=dt, xformers=True)
    inp = synthetic.mv_inputs(frames=w["frames"], pano_hw=w["pano_hw"], pers_hw=w["pers_hw"], seed=1,
                              sam_frames=max(16, w["frames"]), dtype=dt, device=dev)
    cams = synthetic.icosahedron_cameras(90, w["pers_px"], device=dev)
    sch = DDIMScheduler(**configs.NOISE_SCHEDULER_KWARGS)
    sch.set_timesteps(25)
    ts = [torch.tensor([t], dtype=torch.int64, device=dev) for t in sch._timesteps_host]
    pano_in, pers_in = inp["pano_latent"], inp["latents"]
    lat = [pano_in[:1, :4].contiguous(), pers_in[:1, :, :4].contiguous()]

    def step(i):
        pano_in[:, :4] = lat[0]
        pers_in[:, :, :4] = lat[1]
        pp, pn = mv(latents=pers_in, pano_latent=pano_in, timestep=ts[i], prompt_embd=inp["prompt_embd"],
                    pano_prompt_embd=inp["pano_prompt_embd"], cameras=cams, use_fps_condition=True,
                    use_ip_plus_cross_attention=True, fps_tensor_pano=inp["fps_tensor_pano"],
                    fps_tensor_pers=inp["fps_tensor_pers"],
                    reference_images_clip_feat_pano=inp["reference_images_clip_feat_pano"],
                    reference_images_clip_feat_pers=inp["reference_images_clip_feat_pers"],
                    relative_position_tensor=inp["relative_position_tensor"], pitchs_tensor=inp["pitchs_tensor"])
        lat[0] = sch.fused_cfg_step(pn[0:1], pn[1:2], 7.5, sch._timesteps_host[i], lat[0])
        lat[1] = sch.fused_cfg_step(pp[0:1], pp[1:2], 7.5, sch._timesteps_host[i], lat[1])

    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step(2)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            a = agg[ev.name]
            a[0] += 1
            a[1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    ours = sum(v[1] for k, v in rows if "im360" in k)
    blas = sum(v[1] for k, v in rows if "Cijk" in k)
    lines = [f"one eager cfg2 step: {total / 1e3:.1f} ms of device kernels in {sum(v[0] for _, v in rows)} launches; "
             f"im360 kernels {ours / 1e3:.1f} ms, hipBLASLt {blas / 1e3:.1f} ms, other (torch elementwise / copies) "
             f"{(total - ours - blas) / 1e3:.1f} ms"]
    for k, (n, us) in rows[:args.top]:
        lines.append(f"{us / 1e3:9.3f} ms {n:6d} x  {k[:150]}")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:45]))


if __name__ == "__main__":
    main()
