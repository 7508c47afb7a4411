"""The frame-sharded path on the HIP kernels: two ranks (processes) sharing the ONE GPU of the test box, gloo process group
(RCCL refuses two ranks on one device), the all-to-all staged through the host -- everything else is the production path:
``shard_pack`` kernel, the temporal-attention kernel reading the receive buffer in place through its strides, every other
kernel on this rank's frames.  Asserts the frame-sharded ``MultiViewBaseModel.forward`` == the unsharded forward."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret, backend="gloo"):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2, backend="gloo"):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret, backend)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return dict(ret)


def _pack_job(rank, world):
    """shard_pack + exchange + the strided temporal attention on real kernels: sharded == unsharded bit for bit."""
    from imagine360_amd import kernels as K
    from imagine360_amd.dist import FrameShard
    dt, dev = torch.bfloat16, torch.device("cuda", 0)
    b, f, p, heads, d = 2, 16, 203, 8, 40                  # 203 pixels: ragged last pixel shard
    c = heads * d
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(b, f, p, 3 * c, generator=g).to(dev, dt)
    full = K.temporal_attention(qkv.reshape(-1, 3 * c), b, f, p, heads).reshape(b, f, p, c)
    sh = FrameShard(f)
    loc = sh.take(qkv, 1).contiguous()
    q = sh.frames_to_pixels(loc)
    pp = sh.pixels_per_rank(p)
    a = K.temporal_attention(q, b, sh.total, pp, heads, frame_major=True, out=sh.pixel_result_buffer(loc, b, p, c))
    back = sh.pixels_to_frames(a, b, p)
    return bool(torch.equal(back, sh.take(full, 1)))


def test_sharded_temporal_attention_on_the_kernels_is_bit_identical():
    out = _run(_pack_job)
    assert out[0] and out[1]


def _mv_job(rank, world, frames=8, width_div=5, pano_hw=(32, 64), pers_hw=(16, 16), px=128):
    import random
    from imagine360_amd import configs, synthetic as S
    from imagine360_amd.dist import FrameShard, shard_mv_inputs
    dt, dev = torch.float16, torch.device("cuda", 0)
    mv = configs.build_mv_model(width_div, device=dev, dtype=dt, xformers=True)
    mv.noise_on_host = True
    inp = S.cast_mv_inputs(S.mv_inputs(frames=frames, pano_hw=pano_hw, pers_hw=pers_hw, seed=5, sam_frames=max(16, frames)), dev, dt)
    cams = S.icosahedron_cameras(90, px)
    kw = dict(cameras=cams, use_fps_condition=True, use_ip_plus_cross_attention=True)
    torch.manual_seed(3)
    random.seed(3)
    pers_full, pano_full = mv(**kw, **inp)
    pers_full, pano_full = pers_full.cpu(), pano_full.cpu()              # (the big case: both ranks share one GPU)
    torch.cuda.empty_cache()
    sh = FrameShard(frames)
    n_exch = []
    orig = sh.exchange
    sh.exchange = lambda send, tag: (n_exch.append(send.numel() * send.element_size()), orig(send, tag))[1]
    mv.set_frame_shard(sh)
    try:
        torch.manual_seed(3)              # every rank replays the unsharded run's RNG stream (IP noise, WarpAttn coins)
        random.seed(3)
        pers_loc, pano_loc = mv(**kw, **shard_mv_inputs(inp, sh))
    finally:
        mv.set_frame_shard(None)
    pano_loc, pers_loc = pano_loc.cpu(), pers_loc.cpu()
    gathered = sh.gather_frames(pano_loc.contiguous(), 2)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    # worst single frame of this rank's panorama prediction (a frame in the wrong place is an O(1) error of that frame)
    want = sh.take(pano_full, 2)
    per_frame = ((pano_loc.float() - want.float()).pow(2).sum(dim=(0, 1, 3, 4)).sqrt() / want.float().pow(2).sum(dim=(0, 1, 3, 4)).sqrt())
    return [rel(pano_loc, want), rel(pers_loc, sh.take(pers_full, 3)), rel(gathered, pano_full),
            list(pano_loc.shape), bool(torch.isfinite(pano_loc.float()).all()), len(n_exch), sum(n_exch), float(per_frame.max())]


def test_frame_sharded_mv_forward_on_the_kernels_matches_unsharded():
    """Same kernels on the same per-frame data in fp16: the only arithmetic that can differ is hipBLASLt choosing another
    solution for the halved token counts of the small Linears -- last-bit differences that the ~60 layers decorrelate up
    to the storage error of the format (2.1e-3 for one run against fp32, ~3e-3 between two runs); a frame, pixel shard or
    head in the wrong place is an O(1) error."""
    out = _run(_mv_job)
    print("sharded vs unsharded (pano, pers, gathered pano):", {r: out[r][:3] for r in out})
    for r in range(2):
        assert out[r][4] and out[r][3] == [2, 4, 4, 32, 64], out[r]
        assert out[r][0] < 6e-3 and out[r][1] < 6e-3 and out[r][2] < 6e-3, out[r]
        assert out[r][5] == 64, out[r]           # 32 motion modules x (one all-to-all behind the GroupNorm + one in front of the residual add)


def _mv_job_cfg4(rank, world):
    return _mv_job(rank, world, frames=48, width_div=1, pano_hw=(64, 128), pers_hw=(32, 32), px=256)


def test_frame_sharded_whole_forward_at_cfg4_size_matches_unsharded():
    """BASELINE cfg4 itself -- FULL width, 48 frames of 512 x 1024 (64 x 128 panorama latent + 20 views of 32 x 32) -- cut into two
    24-frame shards on two processes: every rank's frames of the frame-sharded forward == the same frames of the unsharded
    48-frame forward (a whole-step invariant at this size that needs no CPU reference).  Exercises, at full
    size: the 17 - 64 frame temporal-attention kernel over frame-major rows, frame positional-encoding rows 0 .. 47 on
    pixel-sharded tokens, shard_pack / unpack of 419 MB slabs, the exchange at the module boundary (64 all-to-alls per step).
    fp16 so that the comparison resolves 6e-3 (the two evaluations differ by hipBLASLt's solution choice for the halved token
    counts and by one extra 16-bit rounding in front of each motion module's residual add)."""
    out = _run(_mv_job_cfg4)
    print("cfg4 sharded vs unsharded (pano, pers, gathered pano, worst frame):", {r: out[r][:3] + [out[r][7]] for r in out})
    # send buffer of one exchange = [W = 2, 24 local frames, images, pixels / 2, C] 16-bit elements; two exchanges per motion module
    expect_bytes = sum(n * 2 * (2 * 24 * imgs * (((hw[0] >> lvl) * (hw[1] >> lvl)) // 2) * c * 2)
                       for imgs, hw in ((2, (64, 128)), (40, (32, 32))) for lvl, (c, n) in enumerate(((320, 5), (640, 5), (1280, 5), (1280, 1))))
    for r in range(2):
        assert out[r][4] and out[r][3] == [2, 4, 24, 64, 128], out[r]
        assert out[r][0] < 6e-3 and out[r][1] < 6e-3 and out[r][2] < 6e-3 and out[r][7] < 1.2e-2, out[r]
        assert out[r][5] == 64 and out[r][6] == expect_bytes, (out[r], expect_bytes)


# ---- RCCL on the one GPU of the test box: backend "nccl" (= RCCL on ROCm) at world size 1 with FrameShard's
#      single-rank shortcuts switched off, so that pack kernel -> all_to_all_single -> temporal kernel on the receive buffer -> return
#      trip, all_gather of frames and the CFG pair exchange go through RCCL device buffers -- eagerly AND captured in a hipGraph, and
#      with two communicators progressing on two streams inside one captured graph (the --dual-stream-shard layout).
def _rccl_job(rank, world, capture_two=False):
    from imagine360_amd import kernels as K
    from imagine360_amd.dist import FrameShard, exchange_cfg_halves, frame_shard_pair
    assert dist.get_backend() == "nccl"
    dt, dev = torch.bfloat16, torch.device("cuda", 0)
    b, f, p, heads, d = 2, 16, 203, 8, 40
    c = heads * d
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(b, f, p, 3 * c, generator=g).to(dev, dt)
    qkv2 = torch.randn(b, f, p, 3 * c, generator=g).to(dev, dt)
    full = K.temporal_attention(qkv.reshape(-1, 3 * c), b, f, p, heads).reshape(b, f, p, c)
    full2 = K.temporal_attention(qkv2.reshape(-1, 3 * c), b, f, p, heads).reshape(b, f, p, c)
    out = {}

    def round_trip(sh, x):
        q = sh.frames_to_pixels(x)
        a = K.temporal_attention(q, b, sh.total, sh.pixels_per_rank(p), heads, frame_major=True, out=sh.pixel_result_buffer(x, b, p, c))
        return sh.pixels_to_frames(a, b, p)

    # 1. eager, through RCCL
    sh = FrameShard(f, force_collectives=True)
    n_exch = []
    orig = sh.exchange
    sh.exchange = lambda send, tag: (n_exch.append(tag), orig(send, tag))[1]
    out["eager"] = bool(torch.equal(round_trip(sh, qkv), full))
    out["exchanges"] = list(n_exch)
    out["gather"] = bool(torch.equal(sh.gather_frames(full.contiguous(), 1), full))
    pair = dist.new_group([0])
    out["cfg_pair"] = bool(torch.equal(exchange_cfg_halves(full, pair), full))
    torch.cuda.synchronize()

    # 2. the same exchange captured in a hipGraph and replayed on new data
    static_in = qkv.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        round_trip(sh, static_in)                      # warm-up on the capture stream (buffers, communicator)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):          # (global mode forbids the NCCL watchdog thread's event queries during the capture: a flaky abort)
        static_out = round_trip(sh, static_in)
    res = []
    for src, want in ((qkv2, full2), (qkv, full)):
        static_in.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        res.append(bool(torch.equal(static_out, want)))
    out["graph"] = res

    # 3. two communicators on two streams inside ONE captured graph
    sh_a, sh_b = frame_shard_pair(f, force_collectives=True)
    in_a, in_b = qkv.clone(), qkv2.clone()
    s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):                                 # warm-up, eagerly on the two streams
        s_main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_main):
            s_side.wait_stream(s_main)
            with torch.cuda.stream(s_side):
                wb = round_trip(sh_b, in_b)
            wa = round_trip(sh_a, in_a)
            s_main.wait_stream(s_side)
        torch.cuda.current_stream().wait_stream(s_main)
        torch.cuda.synchronize()
    out["two_comm_eager"] = [bool(torch.equal(wa, full)), bool(torch.equal(wb, full2))]
    if not capture_two:
        return out
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=s_main, capture_error_mode="thread_local"):
        s_side.wait_stream(s_main)
        with torch.cuda.stream(s_side):
            out_b = round_trip(sh_b, in_b)
        out_a = round_trip(sh_a, in_a)
        s_main.wait_stream(s_side)
    in_a.copy_(qkv2)
    in_b.copy_(qkv)
    graph2.replay()
    torch.cuda.synchronize()
    out["two_comm_graph"] = [bool(torch.equal(out_a, full2)), bool(torch.equal(out_b, full))]
    return out


def _rccl_job_two_comm_capture(rank, world):
    return _rccl_job(rank, world, capture_two=True)


def test_exchange_path_through_rccl_at_world_size_one_eager_and_captured():
    out = _run(_rccl_job, world=1, backend="nccl")[0]
    print("RCCL world size 1:", out)
    assert out["eager"] and out["gather"] and out["cfg_pair"], out
    assert out["exchanges"] == ["f2p_recv", "p2f_recv"], out            # both trips went through all_to_all_single
    assert out["graph"] == [True, True], out
    assert out["two_comm_eager"] == [True, True], out


@pytest.mark.xfail(reason="round 6, ROCm 7.2 / RCCL of this image: ending the capture of a graph in which TWO communicators issue "
                          "collectives on two forked streams segfaults in hipStreamEndCapture (torch.cuda.graphs.capture_end) -- the "
                          "--dual-stream-shard layout stays opt-in and flagged validated_on_rccl: false", strict=False)
def test_two_communicators_on_two_streams_in_one_captured_graph():
    out = _run(_rccl_job_two_comm_capture, world=1, backend="nccl")[0]
    assert out["two_comm_graph"] == [True, True], out
