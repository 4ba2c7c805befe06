#!/usr/bin/env python3
"""Headline benchmark: Mpix/s of rasterizer forward+backward at 3 M Gaussians, SH degree 3, 1297x840.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (preprocess -> depth sort -> scan -> keys -> tile sort -> ranges ->
render forward -> render backward -> preprocess backward) over one synthetic frame per GPU
(SURVEY.md 8d generator, config 3), inputs already resident in HBM.  With N > 1 every rank renders its own
camera view of the same scene (weak scaling) and the step includes the exchange: --parallel allreduce =
Gaussians replicated, per-Gaussian gradients summed by a bucketed RCCL all-reduce overlapped with
preprocess_bwd (the pattern BASELINE.json names); --parallel exchange = Gaussians sharded, splat records and
gradient rows exchanged by all-to-all; the default (auto) times a few steps of both, runs the faster and
reports both.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed live with HIP events
on the launch stream; `cpu_baseline` is the CPU oracle (oracle/gs_oracle.c, a port -- the reference
is a Python loop that needs ~100 min for this frame) timed on this box's host cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (guides/MI355X_MICROARCH.md)

CONFIGS = {
    # name: (N, K, H, W)
    "config3_garden_3M_sh3_1297x840": (3_000_000, 16, 840, 1297),
    "config2_lego_300k_sh3_800x800": (300_000, 16, 800, 800),
    "config4_one_view_6M_sh3_1297x840": (6_000_000, 16, 840, 1297),
    "config4_8views_6M_sh3_1297x840": (6_000_000, 16, 840, 1297),      # BASELINE.json config 4: 8 camera views per step
    "config5_4k_10M_sh3_forward_only": (10_000_000, 16, 2160, 3840),
    "tiny_20k_sh3_320x240": (20_000, 16, 240, 320),
}


def bytes_v1(N, V, D, P, K, tiles, passes):
    """SURVEY.md 8d 'algorithmic bytes per frame', per stage."""
    inb = 4 * (3 + 3 + 4 + 1 + 3 * K)
    return {
        "preprocess_fwd": 40 * N + (inb - 40) * V + 60 * V + 4 * N,
        "scan": 8 * N,
        "duplicate_with_keys": 24 * V + 12 * D,
        "radix_sort": 8 * D + passes * 24 * D,
        "identify_ranges": 8 * D + 8 * tiles,
        "render_fwd": 60 * D + 28 * P,
        "render_bwd": 60 * D + 32 * P + 40 * D,
        "preprocess_bwd": (40 + inb + 60) * V + inb * N,
    }


COUNTERS_NOTE = [None]


def csrc_sha():
    """Hash of the kernel sources the in-tree library is built from (the stamp of profiles/counters.json; the library
    reports the hash it was compiled from as gs_source_sha())."""
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting_amd", "csrc"))
    from source_sha import source_sha
    return source_sha()


def load_counters(config, t_eps):
    """profiles/counters.json if it was measured on these kernel sources, this configuration and stop threshold."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "counters.json")))
    except (OSError, ValueError):
        COUNTERS_NOTE[0] = "profiles/counters.json missing"
        return None
    if os.environ.get("GSPLAT_LIB") or os.environ.get("GSPLAT_BWD_FORM"):
        COUNTERS_NOTE[0] = "a variant library / kernel form is in use"
        return None
    if c.get("csrc_sha") != csrc_sha():
        COUNTERS_NOTE[0] = f"profiles/counters.json was measured on other kernel sources ({c.get('csrc_sha')} != {csrc_sha()})"
        return None
    if c.get("config") != config or abs(float(c.get("t_eps", -1)) - t_eps) > 1e-12:
        COUNTERS_NOTE[0] = "profiles/counters.json was measured on another configuration"
        return None
    return c


def bytes_depth_sort(N):
    """Per-Gaussian depth pre-sort (not in the v1 model: it replaces 4 of radix_sort's 6 passes over D
    by 4 passes over N): 4 B histogram read + 8 B read + 8 B written per Gaussian and pass."""
    return 4 * 20 * N


def bytes_tile_sort(D, tile_bits):
    """The tile sort this build runs on the pairs (the depth order is already in place): per 8-bit pass the key read for the
    histogram (4 B) and the pair read and written (8 + 8 B)."""
    return ((tile_bits + 7) // 8) * 20 * D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="config3_garden_3M_sh3_1297x840", choices=list(CONFIGS))
    ap.add_argument("--t-eps", type=float, default=None, help="per-pixel stop threshold (default: library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-mode", action="store_true",
                    help="skip the extra t_eps = 0 timing (profiling runs: keeps one kind of render launch in the trace)")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--parallel", default="auto", choices=["auto", "allreduce", "exchange"],
                    help="N > 1 training step: 'allreduce' = views sharded, Gaussians replicated, bucketed gradient "
                         "all-reduce overlapped with preprocess_bwd (parallel.ViewShardedRasterizer: the pattern "
                         "BASELINE.json names); 'exchange' = Gaussians sharded, splat records / gradient rows exchanged "
                         "(parallel.SplatExchange); 'auto' = time a few steps of both, run the faster, report both")
    ap.add_argument("--buckets", type=int, default=4, help="allreduce: Gaussian ranges per backward (one all-reduce each)")
    ap.add_argument("--views-per-step", type=int, default=None,
                    help="camera views per training step, ALL ranks together (BASELINE.json config 4: 8); rank r renders "
                         "views r, r + N, ... and the step has ONE gradient exchange (parallel.ViewShardedStep).  Default: "
                         "1 on one GPU (8 for the config4_8views workload); with N > 1, 8 in the all-reduce mode "
                         "(strong scaling of config 4's step) and N in the exchange mode (one view per rank)")
    ap.add_argument("--no-next-rows", action="store_true",
                    help="skip the loss / optimizer / PLY / k-NN legs (counter runs: only the path's kernels in the trace)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N > 1 code path with several ranks on ONE GPU (timings meaningless)")
    ap.add_argument("--forward-only", action="store_true", help="render only (implied by the 4K config)")
    ap.add_argument("--watchdog", type=float, default=0.0,
                    help="seconds after which a still-running bench dumps every thread's stack to stderr and exits")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code path (process group, collectives) even with one rank")
    args = ap.parse_args()

    # stdout carries ONE JSON line: everything libraries print there (RCCL's version banner, gloo's connection
    # notes) goes to stderr instead; the JSON line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.watchdog == 0 and world > 1:
        args.watchdog = 900.0      # a collective that never returns ends with every thread's stack on stderr
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist  # noqa: PLW0621
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import diff_gaussian_rasterization as dgr
    impl = importlib.import_module("gaussian-splatting_amd")
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    prof = importlib.import_module("gaussian-splatting_amd.profiling")
    par = importlib.import_module("gaussian-splatting_amd.parallel")

    N, K, H, W = CONFIGS[args.config]
    t_eps = impl.DEFAULT_T_EPS if args.t_eps is None else args.t_eps
    scene = syn.make_scene(N, K, H, W, seed=0)
    forward_only = "forward_only" in args.config or args.forward_only
    names = ("means3D", "shs", "opacities", "scales", "rotations")

    def make_settings(view):
        return dgr.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=view.tanfovx, tanfovy=view.tanfovy, bg=view.bg.to(dev),
            scale_modifier=1.0, viewmatrix=view.viewmatrix.to(dev), projmatrix=view.projmatrix.to(dev),
            sh_degree=view.sh_degree, campos=view.campos.to(dev), prefiltered=False, debug=False)

    # N > 1: rank r renders view r of the SAME scene (camera jittered per view), one view per rank and step
    # (forward-only 4K frames shard tile rows of ONE view instead: every rank gets the same camera)
    vps = args.views_per_step
    if vps is None and "8views" in args.config:
        vps = 8
    if vps is not None and (vps < world or forward_only):
        raise SystemExit("--views-per-step: need at least one view per rank, and a training configuration")
    views = [syn.jitter_view(scene, r) if (multi and not forward_only) else scene for r in range(world)]
    settings = make_settings(views[rank])
    step_views = 1            # views per step of the mode that runs, all ranks together (set by setup())
    G = scene.G.to(dev)
    rast = dgr.GaussianRasterizer(settings, t_eps=t_eps)
    params, all_settings, ex, vs_rast = {}, None, None, None
    mode = "single"          # single | tile_rows | allreduce | exchange

    my_views, my_G = [], []

    def setup(new_mode):
        nonlocal params, all_settings, ex, vs_rast, mode, step_views, my_views, my_G
        params, all_settings, ex, vs_rast, mode = {}, None, None, None, new_mode
        torch.cuda.empty_cache()
        step_views = 1 if mode in ("tile_rows",) else world
        if mode == "exchange":      # this rank owns Gaussians [lo, hi)
            if vps not in (None, world):
                raise SystemExit("--parallel exchange renders one view per rank and step")
            lo, hi = par.tile_row_band(N, rank, world)
            params = {k: getattr(scene, k)[lo:hi].to(dev).requires_grad_(True) for k in names}
            all_settings = [make_settings(v) for v in views]
            ex = par.SplatExchange(hi - lo, device=dev)
        else:
            params = {k: getattr(scene, k).to(dev).requires_grad_(True) for k in names}
            # several views per step: config 4's step (8 views, all ranks together) by default once there is an exchange
            # to amortise; a single GPU renders one frame per step unless asked otherwise
            if mode != "tile_rows":     # (tile-row sharding cuts ONE frame: step_views stays 1)
                step_views = vps if vps is not None else (8 if (mode == "allreduce" and world > 1 and 8 >= world) else world)
            if mode == "allreduce" or step_views > world:
                my_ids = par.shard_views(step_views, rank, world)
                my_views = [make_settings(syn.jitter_view(scene, v) if (multi or step_views > 1) else scene) for v in my_ids]
                gen = torch.Generator().manual_seed(77)
                my_G = [G] + [(torch.rand(3, H, W, generator=gen) * 2 - 1).to(dev) for _ in my_ids[1:]]
                vs_rast = par.ViewShardedStep(my_views, t_eps=t_eps, num_buckets=args.buckets,
                                              force_collectives=args.force_dist and mode == "allreduce")

    def step():
        if mode == "tile_rows":
            # large-frame inference: this rank renders its band of tile rows, rank 0 assembles the frame
            with torch.no_grad():
                band = par.render_band(settings, rank, world, t_eps=t_eps, means3D=params["means3D"], shs=params["shs"],
                                       opacities=params["opacities"], scales=params["scales"],
                                       rotations=params["rotations"])[0]
                return par.gather_bands(band, H, dst=0)
        if forward_only:
            with torch.no_grad():
                return rast(means3D=params["means3D"], means2D=None, shs=params["shs"], colors_precomp=None,
                            opacities=params["opacities"], scales=params["scales"],
                            rotations=params["rotations"], cov3D_precomp=None)[1]
        for p in params.values():
            p.grad = None
        if mode == "exchange":
            means2D = torch.zeros((world, params["means3D"].shape[0], 3), device=dev, requires_grad=True)
            image, radii, depth, alpha = ex.rasterize(all_settings, params["means3D"], means2D=means2D,
                                                      shs=params["shs"], opacities=params["opacities"],
                                                      scales=params["scales"], rotations=params["rotations"],
                                                      t_eps=t_eps)
            image.backward(G)
            return radii
        if vs_rast is not None:      # this rank's views of the step as one autograd node; allreduce: gradients arrive summed
            means2D = torch.zeros((len(my_views),) + tuple(params["means3D"].shape), device=dev, requires_grad=True)
            images, radii, _, _ = vs_rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                          colors_precomp=None, opacities=params["opacities"],
                                          scales=params["scales"], rotations=params["rotations"], cov3D_precomp=None)
            torch.autograd.backward(images, my_G)
            return radii[0]
        means2D = torch.zeros_like(params["means3D"], requires_grad=True)
        image, radii, depth, alpha = rast(means3D=params["means3D"], means2D=means2D, shs=params["shs"],
                                          colors_precomp=None, opacities=params["opacities"],
                                          scales=params["scales"], rotations=params["rotations"],
                                          cov3D_precomp=None)
        image.backward(G)
        return radii

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    probe = None
    if not multi:
        setup("single")
    elif forward_only:
        setup("tile_rows")
    elif args.parallel != "auto":
        setup(args.parallel)          # an explicit choice that fails ends the run with a non-zero exit status
    else:
        # N > 1 over xGMI has never been run by the builder (one-GPU boxes): time a few steps of both partitionings,
        # run the timed region on the faster, report both.  A mode that raises is reported as such; if both do, exit.
        probe, per_view, views_of = {}, {}, {}
        for m in ("allreduce", "exchange"):
            if m == "exchange" and vps not in (None, world):
                probe[m] = "not applicable (one view per rank and step)"
                continue
            ok = 1
            try:
                setup(m)
                for _ in range(3):
                    step()
                fence()
                t0 = time.perf_counter()
                for _ in range(5):
                    step()
                fence()
                ms = (time.perf_counter() - t0) / 5 * 1e3
            except Exception as e:  # noqa: BLE001
                print(f"[bench] rank {rank}: mode {m} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                ok, ms = 0, float("inf")
            t = torch.tensor([float(ok), -ms if ok else -1e30], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)            # all ranks ok, and the slowest rank's time
            if int(t[0].item()) == 1:
                probe[m] = round(-float(t[1].item()), 4)
                per_view[m] = probe[m] / step_views             # the modes render different numbers of views per step
                views_of[m] = step_views
            else:
                probe[m] = "failed"
        if not per_view:
            raise SystemExit("[bench] both multi-GPU modes failed (see stderr of the ranks)")
        probe["views_per_step"] = views_of
        setup(min(per_view, key=per_view.get))
    exchange = mode == "exchange"
    for _ in range(args.warmup):
        step()
    fence()
    # Untimed pass with HIP events around every stage: finds the dominant kernel and gives the
    # per-stage breakdown.  (Two event records take ~10 us of stream time per stage, ~3 % of a
    # frame, so the timed region below carries events around the dominant kernel only.)
    prof.enable()
    for _ in range(3):
        step()
    stage_ms, stage_calls = prof.disable_and_collect(with_counts=True)
    calls_per_step = {k: v / 3.0 for k, v in stage_calls.items()}
    # the roofline line is about the path's dominant KERNEL; exchange stages (stream waits on RCCL) are reported
    # in stage_ms but are not candidates
    kernel_stages = ("render_bwd", "render_fwd", "preprocess_fwd", "preprocess_bwd", "sort_pairs", "sort_depth",
                     "duplicate_with_keys", "scan_tiles", "identify_tile_ranges")
    dominant = max((k for k in stage_ms if k in kernel_stages), key=stage_ms.get)
    fence()
    prof.enable(only=[dominant])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        radii = step()
    fence()
    dt = time.perf_counter() - t0
    stage_ms[dominant] = prof.disable_and_collect()[dominant]   # mean ms per launch over the timed region
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    frames_per_step = step_views      # tile-row sharding splits ONE frame; a training step renders step_views views
    value = frames_per_step * W * H / (dt / args.steps) / 1e6

    if rank == 0:
        # workload statistics for the byte model
        with torch.no_grad():
            full = {k: getattr(scene, k).to(dev) for k in names} if exchange else \
                {k: params[k].detach() for k in names}
            st = impl.preprocess_and_bin(settings, full["means3D"], full["opacities"], full["scales"],
                                         full["rotations"], None, full["shs"], None)
            V = int((st.radii > 0).sum().item())
            D = st.num_pairs
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        views_per_rank = len(my_views) if vs_rast is not None else 1
        frame_ms = ms_per_step / max(1, views_per_rank)
        lib_sha = importlib.import_module("gaussian-splatting_amd._lib").load().gs_source_sha().decode()
        passes = (32 + max(1, (tiles - 1).bit_length()) + 7) // 8
        b = bytes_v1(N, V, D, W * H, K, tiles, passes)
        stage_of = {"render_bwd": "render_bwd", "render_fwd": "render_fwd", "preprocess_fwd": "preprocess_fwd",
                    "preprocess_bwd": "preprocess_bwd", "sort_pairs": "radix_sort", "sort_depth": "depth_sort",
                    "duplicate_with_keys": "duplicate_with_keys", "scan_tiles": "scan",
                    "identify_tile_ranges": "identify_ranges"}
        if forward_only:
            b["render_bwd"] = b["preprocess_bwd"] = 0
        frame_bytes = sum(b.values())          # SURVEY.md 8d v1 model (the contract figure): one 6-pass sort of 64-bit keys over D
        # The factored sort moves fewer bytes than the v1 row charges: 4 passes over the N depth keys + 2 passes over the D
        # tile ids instead of 6 passes of 24 B over D.  Each launch is priced on the bytes IT moves (so no stage can come
        # out above the HBM peak), and the frame is reported against both totals.
        tile_bits = max(1, (tiles - 1).bit_length())
        b_own = dict(b)
        b_own["radix_sort"] = bytes_tile_sort(D, tile_bits)
        b_own["depth_sort"] = bytes_depth_sort(N)
        b_own["identify_ranges"] = 4 * D + 8 * tiles      # the tile ids are 4-byte keys here (v1 prices 8-byte keys)
        # A multi-view step runs the per-Gaussian stages of its k views as ONE launch each (gs_preprocess_fwd_views /
        # gs_preprocess_bwd_views): the inputs are read and the gradients written once, only the per-view rows k times.
        k_fused = views_per_rank if (vs_rast is not None and views_per_rank > 1) else 1
        fused = {"preprocess_fwd": b["preprocess_fwd"] + (k_fused - 1) * (60 * V + 4 * N),
                 "preprocess_bwd": b["preprocess_bwd"] + (k_fused - 1) * 60 * V}
        if k_fused > 1 and not forward_only:
            b_own["preprocess_fwd"] = fused["preprocess_fwd"] // k_fused      # per view, like every other row
            b_own["preprocess_bwd"] = fused["preprocess_bwd"] // k_fused
        frame_bytes_own = sum(b_own.values())
        b = b_own

        def launch_bytes(st_name):
            """bytes ONE launch of a stage moves: a stage launched once per Gaussian range (the all-reduce mode's
            preprocess_bwd) moves its share of the bytes per launch; a fused multi-view launch moves all its views'"""
            launches = max(calls_per_step.get(st_name, views_per_rank), 1e-9)
            if k_fused > 1 and st_name in fused:
                return int(fused[st_name] / max(1.0, launches))
            return int(b[stage_of[st_name]] / max(1.0, launches / max(1, views_per_rank)))

        dom_bytes = launch_bytes(dominant)
        achieved = dom_bytes / (stage_ms[dominant] * 1e-3) / 1e9
        # every stage of the path, timed by HIP events in THIS run (dominant: mean over the timed region; the
        # others: mean over the three instrumented steps before it); algorithmic bytes from the v1 model
        table = {}
        for st_name, ms in sorted(stage_ms.items(), key=lambda kv: -kv[1]):
            if st_name in stage_of and not (forward_only and st_name.endswith("_bwd")):
                nb = launch_bytes(st_name)
                table[st_name] = {"algorithmic_bytes": nb, "ms": round(ms, 4), "GBps": round(nb / (ms * 1e-3) / 1e9, 1),
                                  "frac_of_hbm_peak": round(nb / (ms * 1e-3) / HBM_PEAK, 4)}
        # Hardware counters cannot be read from inside this process: tools/pmc_collect.sh runs this same command under
        # rocprofv3 --pmc (one pass per counter group) and tools/make_counters.py writes profiles/counters.json,
        # stamped with a hash of the kernel sources.  A stamp that does not match the sources this run was built
        # from means the numbers belong to other kernels: they are then withheld (null), not repeated.
        cnt = load_counters(args.config, t_eps)
        ck = (cnt or {}).get("kernels", {}).get(dominant)
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 4),
                    "traffic": ck.get("hbm_traffic_bytes") if ck else None,
                    "traffic_source": cnt["source"] if ck else ("withheld: " + COUNTERS_NOTE[0] if COUNTERS_NOTE[0] else None),
                    "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(stage_ms[dominant], 4),
                    "kernels": table}
        if ck:
            secs = stage_ms[dominant] * 1e-3
            if ck.get("valu_wave_insts") and ck.get("mean_issue_cycles"):
                peak = 256 * 4 * 2.4e9 / ck["mean_issue_cycles"]
                rate = ck["valu_wave_insts"] / secs
                roofline["valu"] = {"achieved_G_wave_insts_per_s": round(rate / 1e9, 1),
                                    "peak_G_wave_insts_per_s": round(peak / 1e9, 1), "frac": round(rate / peak, 4),
                                    "valu_wave_insts_per_launch": ck["valu_wave_insts"],
                                    "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / mean issue cycles of the visit loop's "
                                                 "instruction mix (profiles/r1f_valu_issue_rates.txt)"}
            if ck.get("write_size_bytes") is not None and dominant == "render_bwd":
                roofline["atomics"] = {"write_bytes_per_launch": ck["write_size_bytes"],
                                       "achieved_GBps": round(ck["write_size_bytes"] / secs / 1e9, 1), "peak_GBps": 1300.0,
                                       "frac": round(ck["write_size_bytes"] / secs / 1.3e12, 4),
                                       "note": "render_bwd writes HBM only through its per-(tile, Gaussian) row atomics "
                                               "(WRITE_SIZE); peak = the guide's ~1.3 TB/s float-atomic rate"}
            if ck.get("lds_bank_conflict_cycles") is not None and ck.get("lds_active_cycles"):
                roofline["lds"] = {"bank_conflict_cycles": ck["lds_bank_conflict_cycles"],
                                   "active_cycles": ck["lds_active_cycles"],
                                   "conflict_frac": round(ck["lds_bank_conflict_cycles"] / ck["lds_active_cycles"], 4)}
            if ck.get("valu_busy_frac") is not None:
                roofline["limiter"] = (f"VALU issue: SQ_ACTIVE_INST_VALU x 4 / SIMD cycles = {ck['valu_busy_frac']:.0%} in "
                                       f"{dominant} ({cnt['source']}); HBM accounting kept as the task prescribes")
        if "limiter" not in roofline:
            roofline["limiter"] = "see DESIGN.md section 4 (no counters for this build / configuration)"
        out = {
            "metric": "Mpix/s fwd+bwd @3M Gaussians SH3" if args.config.startswith("config3") else "Mpix/s " + args.config, "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            # strong: the work of a step is fixed as N grows (one 4K frame cut into tile rows; config 4's 8-view step
            # dealt to the ranks); weak: one view per rank and step
            "scaling": "strong" if (multi and (forward_only or (mode == "allreduce" and step_views != world))) else "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.config, "gaussians": N, "sh_coeffs": K, "width": W, "height": H,
                       "drawn": V, "tile_pairs": D, "t_eps": t_eps,
                       "views_per_step": frames_per_step, "views_per_rank": views_per_rank,
                       "parallelism": {"single": "single", "tile_rows": f"tile_rows{world}",
                                       "exchange": f"views{world}+gaussians_sharded{world}(splat exchange)",
                                       "allreduce": f"views{step_views}over{world}ranks+gaussians_replicated(one bucketed grad "
                                                    f"all-reduce per step, overlapped with the views' one preprocess_bwd launch per range, "
                                                    f"{args.buckets} buckets)"}[mode],
                       "parallel_arg": args.parallel},
            "roofline": roofline,
            "frame_hbm": {"algorithmic_bytes_v1": frame_bytes,
                          "frac_of_peak": round(frame_bytes / (frame_ms * 1e-3) / HBM_PEAK, 4),
                          "bytes_this_algorithm": frame_bytes_own,
                          "frac_of_peak_this_algorithm": round(frame_bytes_own / (frame_ms * 1e-3) / HBM_PEAK, 4),
                          "frame_ms": round(frame_ms, 4),
                          "note": "v1 = SURVEY.md 8d's model (6 radix passes of 24 B over the D pairs): the contract figure; "
                                  "this_algorithm = the same model with the sort rows this build runs (4 x 20 B x N depth "
                                  "keys + 2 x 20 B x D tile ids); frame_ms = step time / views this rank renders per step; in a multi-view step the two per-Gaussian stages are priced as the fused launches they are (inputs read and gradients written once per step, per-view rows once per view), divided by the views"},
            "t_eps_guarantee": (
                f"t_eps = {t_eps:g}: the HIP path equals the oracle run at the SAME stop threshold at the stated bars "
                "(image 1e-4, depth 1e-3, every gradient 1e-3 max|grad|) on configs 2-5 at full size "
                "(tests/test_gpu_parity.py::test_config*_vs_oracle, both modes, oracle at t_eps 0) and on 200 adversarial scenes "
                "(tests/test_gpu_fuzz.py::test_fuzz_200_adversarial_scenes_default_stop_threshold); against the reference's "
                "never-stopping compositing the truncation costs at most t_eps in image / alpha, t_eps max|z| in depth and, per "
                "Gaussian, max(2 t_eps, min(1, t_eps / (1 - opacity))) * sum over its box of max|u| in dL/d opacity -- an "
                "ABSOLUTE bound enforced by tests/test_stop_threshold_bound.py; GSPLAT_T_EPS=0 selects the reference's exact mode"),
            "lib_source_sha": lib_sha,
            "stage_ms": {k: round(v, 4) for k, v in sorted(stage_ms.items(), key=lambda kv: -kv[1])},
        }
        if probe is not None:
            out["parallel_probe_ms_per_step"] = probe      # per STEP; the faster mode per VIEW runs the timed region
        if multi or views_per_rank > 1:   # several launches of a stage per step (one per view / bucket): mean per launch above, launches here
            out["stage_launches_per_step"] = {k: round(v, 2) for k, v in calls_per_step.items() if v != 1.0}
            if mode == "allreduce":
                n_f = sum(int(p[0].numel()) for p in params.values())
                out["allreduce_bytes_per_rank_and_step"] = {
                    "payload": 4 * n_f * N, "note": f"{n_f} fp32 gradients per Gaussian, all {N} Gaussians, "
                                                    f"{args.buckets} buckets launched as gs_preprocess_bwd finishes each range"}
            if exchange:
                n_loc = params["means3D"].shape[0]
                out["exchange_bytes_per_rank_and_step"] = {
                    "forward_out": (world - 1) * n_loc * 60, "backward_in": (world - 1) * n_loc * 4 * impl.rasterizer.grad_row_floats(),
                    "note": "60 B (record 48 + depth key 4 + box 8) per local Gaussian and remote view out, "
                            "64 B gradient row per local Gaussian and remote view back"}
        if world == 1 and not forward_only and t_eps > 0 and not args.no_exact_mode:
            # the same frame without early termination (t_eps = 0: the reference's exact semantics, the mode the
            # parity tests pin); reported beside the headline, which runs at the library's default t_eps
            rast_exact = dgr.GaussianRasterizer(settings, t_eps=0.0)

            def exact_step():
                for p in params.values():
                    p.grad = None
                m2 = torch.zeros_like(params["means3D"], requires_grad=True)
                image = rast_exact(means3D=params["means3D"], means2D=m2, shs=params["shs"], colors_precomp=None,
                                   opacities=params["opacities"], scales=params["scales"],
                                   rotations=params["rotations"], cov3D_precomp=None)[0]
                image.backward(G)
            for _ in range(3):
                exact_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                exact_step()
            torch.cuda.synchronize()
            ms_exact = (time.perf_counter() - t1) / 10 * 1e3
            out["exact_mode"] = {"t_eps": 0.0, "ms_per_step": round(ms_exact, 4),
                                 "value": round(W * H / (ms_exact * 1e-3) / 1e6, 2), "unit": "Mpix/s",
                                 "note": "no early termination; 10 steps, wall time"}
        if world == 1 and not forward_only and not args.no_next_rows:
            out["next_rows"] = {"l1_ssim_loss": bench_loss(impl, H, W, dev),
                                "adam_densify_step": bench_adam(impl, params, radii, dev),
                                "ply_unpack": bench_ply(N, K, dev),
                                "nearest_dist2": bench_knn(params["means3D"].detach(), dev),
                                "training_iteration": bench_iteration(impl, rast, scene, dev)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_iteration(impl, rast, scene, dev, iters=10):
    """The rows chained as train.py:129-218 chains them: render -> L1+SSIM loss -> backward -> Adam step with the
    densification statistics, on the bench scene (the rasterizer's inputs stand in for the raw parameters; the SH
    tensor is one (N,K,3) parameter whose gradient the optimizer consumes in place)."""
    names = ("means3D", "shs", "opacities", "scales", "rotations")
    lrs = {"means3D": 0.00016, "shs": 0.0025, "opacities": 0.05, "scales": 0.005, "rotations": 0.001}
    p = {k: torch.nn.Parameter(getattr(scene, k).to(dev).clone()) for k in names}
    opt = impl.FusedAdam([{"params": [p[k]], "lr": lrs[k], "name": k} for k in names], lr=0.0, eps=1e-15)
    N = p["means3D"].shape[0]
    gt = torch.rand(3, scene.H, scene.W, device=dev)
    acc, den, mr = torch.zeros(N, 1, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, device=dev)

    def it():
        opt.zero_grad(set_to_none=True)
        m2d = torch.zeros((N, 3), device=dev, requires_grad=True)
        image, radii, _, _ = rast(means3D=p["means3D"], means2D=m2d, shs=p["shs"], colors_precomp=None,
                                  opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"],
                                  cov3D_precomp=None)
        impl.l1_ssim_loss(image, gt, lambda_dssim=0.2).backward()
        opt.step(viewspace_grad=m2d.grad, radii=radii, xyz_gradient_accum=acc, denom=den, max_radii2D=mr)

    for _ in range(3):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"ms": round(ms, 3), "iterations_per_s": round(1e3 / ms, 1),
            "note": "render + fused L1/SSIM loss + backward + fused Adam/densification statistics, wall time"}


def bench_knn(points, dev, iters=5):
    """SURVEY.md 8f row 4 (outside the headline metric): distCUDA2 on the bench scene's means (the reference's
    N x N torch.cdist cannot run at this size: 3 M points = 36 TB)."""
    knn = importlib.import_module("gaussian-splatting_amd.knn")
    d, st = knn.distCUDA2(points, return_stats=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        knn.distCUDA2(points)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"ms": round(ms, 3), "points": points.shape[0], "Mpoints_per_s": round(points.shape[0] / ms / 1e3, 1),
            "grid": list(st["grid"]), "isolated_points": st["isolated"], "crowded_points": st["crowded"],
            "note": "wall time incl. the bounding-box read-back; exact 1-NN, O(N) memory"}


def bench_ply(N, K, dev, iters=10):
    """SURVEY.md 8f row 3 (outside the headline metric): unpack N vertex records of a trained-model PLY (already in
    HBM) into the six parameter tensors + the four activated rasterizer inputs.  Algorithmic bytes: the records
    read once, every output written once."""
    import ctypes as C
    lib_mod = importlib.import_module("gaussian-splatting_amd._lib")
    lib = lib_mod.load()
    layout = lib_mod.GsPlyLayout()
    R = 3 * (K - 1)
    lib.gs_ply_write_header(None, 0, N, R, C.byref(layout))
    rec = torch.randn(N * layout.stride_bytes // 4, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    outs = [torch.empty((N, w), **f32) for w in (3, 3, R, 1, 3, 4, R + 3, 1, 3, 4)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib_mod.check(lib.gs_ply_unpack(C.byref(layout), C.c_void_p(rec.data_ptr()),  # noqa: E731
                                                   *[C.c_void_p(o.data_ptr()) for o in outs], st), "gs_ply_unpack")
    for _ in range(2):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    nbytes = rec.numel() * 4 + sum(o.numel() for o in outs) * 4
    return {"ms": round(ms, 4), "algorithmic_bytes": nbytes, "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4)}


def bench_adam(impl, params, radii, dev, iters=10):
    """SURVEY.md 8f row 2 (outside the headline metric): one fused Adam step over the reference's six parameter
    groups of the bench scene + the densification statistics, using the gradients the last frame left in .grad.
    f_dc / f_rest are the column ranges of the SH tensor (their gradients are read in place from the (N,K,3)
    SH gradient).  Algorithmic bytes: 28 per parameter (p, g, m, v read; p, m, v written) + 32 per Gaussian."""
    N, K = params["shs"].shape[0], params["shs"].shape[1]
    g_shs = params["shs"].grad
    leaves = {"xyz": (params["means3D"], params["means3D"].grad, 0.00016),
              "f_dc": (params["shs"].detach()[:, :1].contiguous(), g_shs[:, :1], 0.0025),
              "f_rest": (params["shs"].detach()[:, 1:].contiguous(), g_shs[:, 1:], 0.0025 / 20),
              "opacity": (params["opacities"], params["opacities"].grad, 0.05),
              "scaling": (params["scales"], params["scales"].grad, 0.005),
              "rotation": (params["rotations"], params["rotations"].grad, 0.001)}
    groups, nparam = [], 0
    for name, (p, g, lr) in leaves.items():
        q = torch.nn.Parameter(p.detach().clone())
        q.grad = g
        nparam += q.numel()
        groups.append({"params": [q], "lr": lr, "name": name})
    opt = impl.FusedAdam(groups, lr=0.0, eps=1e-15)
    vg = torch.randn(N, 3, device=dev)
    acc, den, mr = (torch.zeros(N, 1, device=dev), torch.zeros(N, 1, device=dev), torch.zeros(N, device=dev))
    kw = dict(viewspace_grad=vg, radii=radii, xyz_gradient_accum=acc, denom=den, max_radii2D=mr)
    for _ in range(2):
        opt.step(**kw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        opt.step(**kw)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    # the reference's own step on this GPU: torch.optim.Adam (foreach) on contiguous copies of the same gradients
    tg = []
    for grp in groups:
        q = torch.nn.Parameter(grp["params"][0].detach().clone())
        q.grad = grp["params"][0].grad.contiguous()
        tg.append({"params": [q], "lr": grp["lr"], "name": grp["name"]})
    ref = torch.optim.Adam(tg, lr=0.0, eps=1e-15)
    for _ in range(2):
        ref.step()
    a.record()
    for _ in range(iters):
        ref.step()
    b.record()
    torch.cuda.synchronize()
    ms_ref = a.elapsed_time(b) / iters
    nbytes = 28 * nparam + 32 * N
    return {"ms": round(ms, 4), "torch_adam_ms": round(ms_ref, 4), "parameters": nparam, "algorithmic_bytes": nbytes,
            "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4),
            "note": "one launch: 6 groups (reference lrs, eps 1e-15) + max_radii2D / xyz_gradient_accum / denom"}


def bench_loss(impl, H, W, dev, iters=20):
    """SURVEY.md 8f row 1 (outside the headline metric): fused L1 + SSIM loss, value + d/d image, on one
    (3,H,W) frame; HIP events around the call.  Algorithmic bytes: 8 read + 12 written (forward maps),
    20 read + 4 written (backward) per pixel-channel."""
    g = torch.Generator().manual_seed(5)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    img = (gt + 0.05 * torch.randn(3, H, W, generator=g).to(dev)).clamp(0, 1).requires_grad_(True)
    for _ in range(3):
        impl.l1_ssim_loss(img, gt, 0.2).backward()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        img.grad = None
        impl.l1_ssim_loss(img, gt, 0.2).backward()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    nbytes = 44 * 3 * H * W
    return {"ms_fwd_bwd": round(ms, 4), "algorithmic_bytes": nbytes,
            "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4),
            "note": "includes the autograd plumbing (two small torch kernels) around the three HIP launches"}


def cpu_baseline(scene):
    """The CPU oracle (a C port of the reference algorithm) on the same frame, host cores of this box."""
    from oracle import oracle
    oracle.build()
    cores = min(oracle.max_threads(), int(os.environ.get("OMP_NUM_THREADS", "16")))
    oracle.set_threads(cores)
    a = lambda t: t.numpy()  # noqa: E731
    t0 = time.perf_counter()
    oracle.forward_backward(scene.H, scene.W, scene.sh_degree, a(scene.viewmatrix), a(scene.projmatrix),
                            a(scene.campos), a(scene.bg), a(scene.means3D), a(scene.opacities),
                            scales=a(scene.scales), rotations=a(scene.rotations), shs=a(scene.shs),
                            G_C=a(scene.G))
    dt = time.perf_counter() - t0
    return {"value": round(scene.W * scene.H / dt / 1e6, 4), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"one full frame of the same workload ({scene.means3D.shape[0]} Gaussians, "
                      f"{scene.W}x{scene.H}, fwd+bwd), {dt:.2f} s wall",
            "reference_python_loop": "19.48 s for 10k Gaussians 200x200 (0.00205 Mpix/s, BASELINE.md, 8 cores)",
            "torch_loop_port_config1": cpu_baseline_torch_loop()}


def cpu_baseline_torch_loop(n_sample=1500):
    """the build's PyTorch-CPU restatement of the reference's per-Gaussian loop
    (oracle/torch_loop_oracle.py, pinned against the reference's goldens) on a bounded sample of config 1
    (10 k Gaussians, SH degree 0, 200x200): the loop costs the same per Gaussian whatever N is."""
    from oracle import torch_loop_oracle as tl
    syn = importlib.import_module("gaussian-splatting_amd.synthetic")
    sc = syn.make_scene(n_sample, 1, 200, 200, seed=0)
    leaves = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs[:, 0, :], sc.opacities, sc.scales, sc.rotations)]
    t0 = time.perf_counter()
    img, _, _ = tl.render(*leaves, sc.viewmatrix, sc.projmatrix, sc.bg, 200, 200)
    t1 = time.perf_counter()
    (img * sc.G).sum().backward()
    t2 = time.perf_counter()
    per = (t2 - t0) / n_sample
    return {"ms_per_gaussian": round(per * 1e3, 3), "fwd_s": round(t1 - t0, 2), "bwd_s": round(t2 - t1, 2),
            "threads": torch.get_num_threads(), "sample": f"{n_sample} of config 1's 10 000 Gaussians, 200x200, SH degree 0",
            "config1_extrapolated": {"seconds": round(per * 10000, 1), "Mpix_per_s": round(0.04 / (per * 10000), 5),
                                     "note": "linear in the number of Gaussians (SURVEY.md section 6: 1 000 -> 1.9 s, 10 000 -> 19.5 s)"}}


if __name__ == "__main__":
    main()
