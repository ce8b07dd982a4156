#!/usr/bin/env python
"""Benchmark of the WOFT hot path on MI355X: tracked frames/sec at 1080p, 12 RAFT iterations.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N>1: one rank per GPU over RCCL; launched by torch.distributed.run -- by the caller, or by bench.py itself
   when it is started as a plain `python bench.py --gpus N` without the torchrun environment)

One step = one tracker.track() call: pre-warp of the frame, weighted-RAFT flow (full model, 12
iterations) template -> frame, masking + Sobol-500 subsampling, weighted least-squares homography,
re-detection test -- the reference's default configuration (configs/WOFT.py).  Frames are
synthetic, already resident in HBM when the timed region starts; weights are a seeded
synthetic checkpoint with the reference's key set (the trained checkpoints are not in the snapshot).
The sequence is a series of 48-frame clips of SURVEY 8d's motion (with random weights the estimated pose is
meaningless and eventually leaves the image, where the reference's estimator raises on < 4 points; every clip
therefore starts from the template's pose -- the per-frame work is the same for every frame of every clip).
Each rank tracks its own sequence; the finished tracks are all-gathered (RCCL) at the end.

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields).
"""
import argparse
import gc
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

LOOKUP_ALGO_BYTES_PER_PIXEL = 4 * (10 * 10 * 4 + 9 * 9 * 4)        # 2896 B
HBM_PEAK_GBS = 8000.0                                             # MI355X_MICROARCH.md: 8.0 TB/s spec


CLIP = 48      # frames per clip of the synthetic sequence

# flow EPE of the GPU path against the oracle, (mean, max) px at 12 iterations.  fp32 and the fp32-emulating
# bf16x3 share the GPU-fp32 budget; bf16: 0.05 px mean (max: the bf16 tests' 0.25 px); fp16 (mixed_precision scoping):
# the tested budgets of tests/test_flow_gpu.py
EPE_BUDGET = {"fp32": (1e-3, 1e-2), "bf16x3": (1e-3, 1e-2), "bf16": (0.05, 0.25), "fp16": (0.01, 0.05), "f16mx8": (1e-3, 1e-2)}


def restart_clip(tracker):
    """Pose state of a freshly initialised tracker (TRK:43-47); template-side tensors stay as they are."""
    tracker.prev_H2init = np.eye(3)
    tracker.last_good_H2init = np.eye(3)
    tracker.lost, tracker.N_lost = False, 0


def _overrule_run(inner, seen, every, frame, prewarp_H):
    """Global stage with the re-detection verdict overruled on the last two frames of every `every` (bench.py's lost-run pass)."""
    fit = inner(frame, prewarp_H)
    seen["i"] += 1
    if seen["i"] % every in (every - 2, every - 1):
        fit.success = False
    return fit


def make_sequence(H, W, seq_id, n_frames):
    """Template (numpy, host) + frames (CUDA uint8 tensors) warped on the device."""
    from woft_amd import ops, synth
    template = synth.make_template(H, W, seq_id=seq_id)
    tg = torch.from_numpy(template).cuda()
    frames = []
    for t in range(1, min(n_frames, CLIP) + 1):          # (frame i of a longer run is frames[i % CLIP])
        out = torch.empty_like(tg)
        # SURVEY 8d's motion H_t, t = 1 .. CLIP; longer runs are further clips of the same motion (the tracker's pose
        # is put back to the template's at each clip start, restart_clip) so that any --steps keeps the object in view
        ops.warp_perspective_u8(tg, synth.seq_homography(t, H, W), out, None)
        frames.append(out)
    torch.cuda.synchronize()
    return template, frames


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(sd, template, mask, frames_np, iters):
    """The CPU oracle (torch-CPU restatement of the reference path, parity-checked against the imported reference)
    timed on the host cores: a short tracked sequence, one wall time per frame (>= 3 frames, median).
    -> (seconds per frame, dense correspondences of the FIRST frame for the EPE gate)."""
    from oracle import tracker_ref
    ref = tracker_ref.TrackerRef(sd, iters=iters)
    ref.init(template, mask)
    keep = {}
    orig = ref._flow

    def flow(a, b):
        r = orig(a, b)
        keep.setdefault("tc", r)
        return r
    ref._flow = flow
    times = []
    for f in frames_np:
        t0 = time.perf_counter()
        ref.track(f)
        times.append(time.perf_counter() - t0)
    return times, keep["tc"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed track() calls (default 200: a ~2 s timed region; the extra passes use min(steps, 40))")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16x3", "bf16", "fp16", "f16mx8"],
                    help="conv / correlation-GEMM arithmetic: exact fp32 MFMA, split-bf16 (fp32-emulating), bf16, fp16 scoping. "
                         "Default: whatever the SHIPPED flow config selects (pytracking/optical_flow/configs/"
                         "v2_SNOB_large_g05_RAFT.py: the configuration a drop-in user runs) -- reported as config.precision_source")
    ap.add_argument("--corr", default="otf", choices=["volume", "otf"],
                    help="correlation: volume-free on-the-fly lookup (default in the split-bf16 precisions) or the "
                         "all-pairs volume in HBM whose lookup is the HBM-roofline kernel (bit-identical results; "
                         "the exact-fp32 passes are volume-free too since round 3: WOFT_FP32_CORR=volume for the A/B)")
    ap.add_argument("--full-weight-head", action="store_true",
                    help="evaluate the weight head on every template pixel, as the reference's network does, instead of "
                         "only where the tracker reads the weights (its template mask: the tracker's default; identical "
                         "tracks).  The default run times this variant too ('alt_weight_head').")
    ap.add_argument("--no-alt-corr", action="store_true",
                    help="skip the short extra run in the other correlation mode (reported under 'alt_corr'; in the "
                         "default mode it also measures the volume lookup for 'roofline_lookup')")
    ap.add_argument("--no-alt-precisions", action="store_true",
                    help="skip the short extra runs at the other two precisions (reported under 'alt_precisions')")
    ap.add_argument("--tracker-config", default="WOFT", choices=["WOFT", "WOFT_IRLS"],
                    help="reference-format tracker config under pytracking/configs: weighted LSq (the reference's "
                         "default, WOFT.py) or the IRLS estimator (BASELINE config 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ladder", action="store_true",
                    help="skip the 'reference_work' (full weight head in bf16x3 and in exact fp32), 'lost_frame' and "
                         "'steady_state' passes")
    ap.add_argument("--no-template-cache", action="store_true",
                    help="recompute the template's features every frame, as the reference does")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- N ranks of this same command line, one per GPU
        # (LOCAL_RANK -> device), rendezvous on the loopback address; rank 0 prints the JSON line.  On a box with fewer
        # than N GPUs (the 1-GPU test box) the ranks share device 0 and gather over gloo (WOFT_SINGLE_DEVICE).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if torch.cuda.device_count() < args.gpus:
            env["WOFT_SINGLE_DEVICE"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        for line in proc.stdout:                 # stdout carries the ONE JSON line; library chatter goes to stderr
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
        sys.exit(proc.wait())

    from woft_amd import dist as wdist, ops, synth
    from pytracking.utils.config import load_config
    rank, world, local = wdist.init_distributed()
    affinity0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    binding = wdist.bind_to_gpu_node(wdist.device_index(), local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks "
                         f"(or run plain `python bench.py --gpus {args.gpus}`, which launches them itself)")
    H, W, K, Wm = args.height, args.width, args.steps, args.warmup
    assert H % 8 == 0 and W % 8 == 0

    sd = synth.make_state_dict(seed=7)
    template, frames = make_sequence(H, W, rank, CLIP)       # (every pass indexes frames[i % CLIP])
    mask = synth.make_init_mask(H, W)

    def make_tracker(precision, corr=None, mask_wh=None, graph=False, first=None):
        conf = load_config(ROOT / "pytracking" / "configs" / (args.tracker_config + ".py"))
        conf.mask_weight_head = (not args.full_weight_head) if mask_wh is None else mask_wh
        conf.flow_config.model = sd
        conf.flow_config.iters = args.iters
        if precision is not None:                 # (None: the shipped flow config's own `precision` key decides)
            conf.flow_config.precision = precision
        conf.flow_config.graph = graph
        precision = precision or conf.flow_config.precision or "bf16x3"
        # (exact fp32: volume-free since round 3 -- bit-identical to the volume path, no P x P buffer; WOFT_FP32_CORR=volume: A/B)
        conf.flow_config.corr = (corr or args.corr) if precision != "fp32" else (corr or os.environ.get("WOFT_FP32_CORR", "otf"))
        trk = conf.tracker_class(conf)
        trk.init(template if first is None else first, mask)
        if args.no_template_cache:
            trk.flower.pin_source(None)
        return trk

    tracker = make_tracker(args.precision)
    precision_source = ("bench.py --precision" if args.precision else
                        f"{tracker.flower.precision_source} of the shipped pytracking/optical_flow/configs/v2_SNOB_large_g05_RAFT.py")
    args.precision = tracker.flower.precision            # (from here on: the precision the timed tracker really runs)
    plan = tracker.flower.engine.plan(H, W)
    corr_mode = tracker.flower.engine.corr

    EVENT_EVERY = 4        # frames between two frames whose dominant-kernel launches carry HIP events (see below)

    def track_all(trk, first, n, timed=None):
        """Track frames[first : first + n] (0-based index i = frame t - 1; a new clip starts where i % CLIP == 0).
        timed = (plan, wh_events, conv_events): the per-launch HIP events of the roofline fields are recorded on every
        EVENT_EVERY-th frame of the span only -- an event record is a stream barrier packet (kernel trace: 6-8 us of idle
        matrix pipes per timed launch, 38 timed launches per frame = 3 % of the frame time the events are there to explain)."""
        res = []
        for i in range(first, first + n):
            if i > 0 and i % CLIP == 0:
                restart_clip(trk)
            if timed is not None:
                on = (i - first) % EVENT_EVERY == 0
                timed[0].wh_events, timed[0].conv_events = (timed[1], timed[2]) if on else (None, None)
            res.append(trk.track(frames[i % CLIP]))
        return res

    results = track_all(tracker, 0, Wm)
    wdist.gather_tracks(results[:1])        # untimed: creates the RCCL communicator / warms the collective
    torch.cuda.synchronize()
    # the kernel with the largest share of a frame (profiles/r03_bench_kernel_stats_bf16x3.csv): the 3x3 / 64-column
    # instance of the register-streamed-weights conv kernel = the motion encoder's three 3x3 layers (update.py:83,85,86)
    ROOF_TAGS = ("convc2", "convf2", "convc2+convf2", "convm")
    # (HIP events are stream-ordered barriers: the volume-free lookup, not this mode's roofline kernel, is timed in isolation
    #  below -- 'lookup_otf_by_flow_field' -- instead of inside the timed region)
    plan.lookup_events = [] if corr_mode == "volume" else None
    wh_events, conv_events = [], {t: [] for t in ROOF_TAGS}
    wdist.barrier()
    torch.cuda.synchronize()
    wait0 = tracker.host_wait_s
    t0 = time.perf_counter()
    results += track_all(tracker, Wm, K, timed=(plan, wh_events, conv_events))
    t_track = time.perf_counter() - t0
    host_busy = t_track - (tracker.host_wait_s - wait0)   # the rank's Python launch loop: wall time not spent blocked on the GPU
    tracks = wdist.gather_tracks(results[Wm:])
    torch.cuda.synchronize()
    wdist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = wdist.gather_floats([elapsed, binding.get("numa_node") if binding.get("numa_node") is not None else -1,
                                    binding.get("cores") or 0, float(bool(binding.get("bound"))), host_busy])
    elapsed = wdist.max_over_ranks(elapsed)
    events = plan.lookup_events
    plan.lookup_events = plan.wh_events = plan.conv_events = None

    if rank != 0:
        return
    n_lost = int(tracks[:, :, 9].sum().item())
    peak_gb = torch.cuda.max_memory_allocated() / 2 ** 30        # every device buffer of the path is a torch tensor

    mask_region = not args.full_weight_head
    # (matrix-pipe passes per product, in units of one bf16 MFMA pass: f16mx8 = 1 fp16 pass + 2 block-scaled fp8 passes at twice the rate)
    terms = {"fp32": 1, "bf16x3": 3, "bf16": 1, "fp16": 1, "f16mx8": 2}[args.precision]
    mfma_peak = 157.3 if args.precision == "fp32" else 2500.0      # TFLOP/s dense: f32-input MFMA / bf16 MFMA (MI355X_MICROARCH.md)

    def lookup_roofline(evs, P, storage="fp32"):
        """Correlation lookup in the volume (the named HBM-roofline kernel): algorithmic bytes / live time.
        `traffic` = HBM bytes per launch from PMC passes (FETCH_SIZE with the guide's gfx950 correction + WRITE_SIZE), which
        need their own rocprofv3 runs: tools/lookup_pmc.sh regenerates profiles/r03_lookup_pmc.json on the GPU box."""
        lk_ms = [s.elapsed_time(e) for s, e in evs]
        lk_avg = float(np.mean(lk_ms)) if lk_ms else float("nan")
        # (2r+2)^2 volume elements in, (2r+1)^2 fp32 samples out, per level: 2896 B with the fp32 volume, 2096 B with the
        # bf16-storage volume of the plain-bf16 operating point
        algo_bytes = (LOOKUP_ALGO_BYTES_PER_PIXEL if storage == "fp32" else 4 * (10 * 10 * 2 + 9 * 9 * 4)) * P
        traffic, src = None, None
        for name in (("r05_lookup_pmc.json", "r04_lookup_pmc.json", "r03_lookup_pmc.json", "r02_lookup_pmc.json", "r01_lookup_pmc.json") if storage == "fp32" else ()):   # (PMC passes: fp32 volume;
                                                        # FETCH_SIZE is uncalibrated for the bf16 volume's 8-B-per-lane loads)
            try:
                pmc = json.loads((ROOT / "profiles" / name).read_text())
                if pmc["resolution"] == [H, W]:
                    traffic, src = pmc["traffic_bytes_per_launch"], f"profiles/{name} (tools/lookup_pmc.sh: separate rocprofv3 --pmc passes)"
                    break
            except Exception:
                pass
        achieved = algo_bytes / (lk_avg * 1e-3) / 1e9 if lk_ms else float("nan")
        return {"bound": "hbm", "kernel": "corr_lookup_kernel<4>" if storage == "fp32" else "corr_lookup_kernel<4, bf16 volume>",
                "volume_storage": storage, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": lk_avg, "launches_timed": len(lk_ms)}

    def conv_roofline(evs_by_tag, layers):
        """The kernel symbol with the largest share of a frame: conv_regb_kernel<8,16,3,3, 2 x 2 waves> (64-column tiles),
        i.e. the motion encoder's 3x3 convs on the 1/8-resolution map -- convc2 256->192 and convf2 128->64 (independent
        branches: ONE launch, woft_conv2d_pair) and conv 256->126 (update.py:83,85,86,91-96), 24 launches per frame;
        matrix-core bound.  HIP events around ALL its launches in the timed
        region (on the stream the kernels are enqueued on): achieved = sum of the launches' 2*M*K*N / sum of their times, so
        that avg_launch_ms is comparable with the rocprofv3 average of the same symbol."""
        tot_ms, tot_fl, tot_issued, n_l, per = 0.0, 0.0, 0.0, 0, {}
        for tag, ps in layers.items():                  # (a launch may hold two layers: woft_conv2d_pair)
            ms = [s.elapsed_time(e) for s, e in evs_by_tag.get(tag, [])]
            if not ms:
                continue
            fl = sum(2.0 * p._m * p.taps_y * p.taps_x * p.cin_pad * p.cout for p in ps)
            tot_ms += float(np.sum(ms))
            tot_fl += fl * len(ms)
            tot_issued += sum(2.0 * (p._m_tiles * 128) * p.taps_y * p.taps_x * p.cin_pad * p.cout_pad for p in ps) * terms * len(ms)
            n_l += len(ms)
            per[tag] = {"conv": " + ".join(f"3x3 {p.cin_pad}->{p.cout}" for p in ps) + (" in one launch" if len(ps) > 1 else ""),
                        "avg_launch_us": 1e3 * float(np.mean(ms)), "algorithmic_flops_per_launch": fl}
        p0 = next(iter(layers.values()))[0]
        kern = {8: "conv_regb_kernel<8,16,3,3> (weights streamed global -> registers)", 1: "conv_halo_bf16_kernel<8,16,3,3>",
                4: "conv_halo_bf16_kernel<4,16,3,3>", 0: "conv_mfma_f32_kernel"}.get(p0.halo, f"halo {p0.halo}")
        t = tot_ms * 1e-3
        # HBM bytes per launch of this symbol from the committed PMC passes (tools/conv_pmc.sh; 1080p, default precision only)
        traffic, tsrc = None, None
        if p0.halo == 8 and p0._m == 135 * 240 and args.precision == "bf16x3":
            for name in ("r05_conv_pmc.json", "r04_conv_pmc.json", "r03_conv_pmc.json"):
                pth = ROOT / "profiles" / name
                if pth.exists():
                    traffic = json.loads(pth.read_text())["traffic_bytes_per_launch"]
                    tsrc = f"profiles/{name} (tools/conv_pmc.sh: separate rocprofv3 --pmc passes, mean over the symbol's dispatches)"
                    break
        return {"bound": "mfma", "kernel": f"{kern}, tile_n {p0.tile_n}: motion-encoder 3x3 convs on {p0._m} pixels",
                "achieved": tot_fl / t / 1e12, "peak": mfma_peak, "unit": "TFLOP/s", "frac": tot_fl / t / 1e12 / mfma_peak,
                "traffic": traffic, "traffic_source": tsrc, "matrix_core_issue_frac": tot_issued / t / 1e12 / mfma_peak,
                "algorithmic_flops_per_launch": tot_fl / max(n_l, 1), "mfma_terms_per_product": terms,
                "avg_launch_ms": tot_ms / max(n_l, 1), "launches_timed": n_l, "layers": per,
                "note": "frac prices the launches' own 2*M*K*N products against the dense peak of the MFMA type used; the "
                        "issue fraction also counts the 3 bf16 MFMAs per fp32-emulating product and tile padding"}

    def wh_roofline(evs, layer, P):
        """Weight-head 3x3 128->128 layer on 9x9 lookup windows (weighted_raft.py:337-340; the first 5->128 layer is
        computed inside the same launch), matrix-core bound; HIP events around its launches in the timed region."""
        ms = [s.elapsed_time(e) for s, e, _ in evs]
        wh_ms = float(np.mean(ms)) if ms else float("nan")
        n = int(layer.h)
        # windows per launch: P, the mask region, or -- sparse weight head -- the device-side count of windows that really ran
        n_win = float(np.mean([float(k) for _, _, k in evs])) if evs else float(P)
        flops = 2.0 * n_win * n * n * 9 * 128 * 128
        rows = 96 if (n == 9 and args.precision != "fp32") else n * n  # 81 pixels occupy 3 MFMA row tiles
        issued = flops * terms * rows / (n * n)
        if bool(getattr(plan, "wh0_fused", False)):
            flops += 2.0 * n_win * n * n * 45 * 128
            issued += 2.0 * n_win * 96 * 48 * 128 * terms            # K 45 -> 48
        return {"bound": "mfma", "kernel": "weight head conv 3x3 128->128 (+ fused 5->128 first layer) on 9x9 windows",
                "achieved": flops / (wh_ms * 1e-3) / 1e12, "peak": mfma_peak, "unit": "TFLOP/s",
                "frac": flops / (wh_ms * 1e-3) / 1e12 / mfma_peak, "traffic": None,
                "matrix_core_issue_frac": issued / (wh_ms * 1e-3) / 1e12 / mfma_peak,
                "algorithmic_flops_per_launch": flops, "windows_per_launch": n_win, "mfma_terms_per_product": terms,
                "avg_launch_ms": wh_ms, "launches_timed": len(ms)}

    sparse_wh = bool(getattr(tracker, "_sparse_weights", False))
    wh_desc = (("the windows under the upsampling support of the <= 500 Sobol-drawn correspondences the fit reads (the draw is "
                "decided by the flow alone; TRK:287-312 + subsampler), inside the template-mask region" if sparse_wh else
                "1/8-res pixels of the template mask (N_in = HW/4, SURVEY 8d) + upsampling support: the weights the tracker "
                "reads (TRK:287-312)") + ": the tracker's default; identical tracks to evaluating it everywhere (checked below)"
               if mask_region else "every template pixel (as the reference's network evaluates it)")
    out = {
        "metric": "tracked frames/sec at 1080p, 12 RAFT iters; flow EPE vs reference",
        "value": world * K / elapsed, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": 1000.0 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA operands emulating fp32, fp32 accumulate)",
                  "bf16": "bf16 (fp32 accumulate)",
                  "fp16": "fp16 convolutions (fp32 accumulate) in the encoders and the update block, bf16x3 correlation and "
                          "weight head: the reference's mixed_precision scoping",
                  "f16mx8": "fp16 x fp16 + two block-scaled fp8 cross terms per product (fp32-emulating in 2 matrix-pipe passes, fp32 "
                            "accumulate) in the update block's convolutions; bf16x3 elsewhere"}[args.precision], "data": "synthetic",
        "config": {"workload": f"{H}x{W} synthetic sequence per GPU ({CLIP}-frame clips): WeightedRAFT-full {args.iters} iters + "
                               + ("weighted LSq homography on Sobol-500 correspondences (reference default config WOFT.py)"
                                if args.tracker_config == "WOFT" else
                                "IRLS (Huber) homography on Sobol-500 correspondences (reference config WOFT_IRLS.py)"),
                   "resolution": [H, W], "iters": args.iters, "sequences": world, "correlation": corr_mode,
                   "weight_head": wh_desc,
                   "template_cache": not args.no_template_cache, "weights": "synthetic seed 7 (reference key set)",
                   "frames_resident_in_hbm": True, "precision": args.precision, "precision_source": precision_source,
                   "update_block": "one launch per layer (9 launches per refinement iteration)",
                   "solver": getattr(tracker, "solver_decision", None)},
        "lost_frames": n_lost, "hbm_allocated_peak_gb": peak_gb, "tracks_gathered": [int(tracks.shape[0]), int(tracks.shape[1])],
        # one entry per rank (a straggler shows here; `value` uses the slowest rank): ms per step inside the same barriers,
        # the NUMA node of the rank's GPU and the host cores its launch thread is pinned to (woft_amd.dist.bind_to_gpu_node)
        "per_rank": [{"rank": r, "ms_per_step": 1000.0 * float(v[0]) / K, "gpu_numa_node": (int(v[1]) if v[1] >= 0 else None),
                      "host_cores": int(v[2]), "bound": bool(v[3]),
                      # host time of the rank's launch loop per frame (track() wall time minus the time blocked on the per-flow
                      # result read): what must stay well below the GPU's frame time for N loops on two sockets not to limit N GPUs
                      "host_busy_ms_per_step": 1000.0 * float(v[4]) / K} for r, v in enumerate(per_rank.tolist())],
    }
    out["host_busy_ms_per_step"] = max(r["host_busy_ms_per_step"] for r in out["per_rank"])
    layers = {e[2]: (list(e[1]) if e[0] == "conv2" else [e[1]]) for e in plan.prog_iter if len(e) > 2 and e[2] in ROOF_TAGS}
    same = {(p_.halo, p_.tile_n, p_.taps_y, p_.taps_x) for ps_ in layers.values() for p_ in ps_}
    if len(same) > 1:                    # (another resolution / precision picked different kernels: keep the largest layer)
        layers = {k_: v_ for k_, v_ in layers.items() if k_.startswith("convc2")}
    have_conv = bool(layers) and any(conv_events.get(t) for t in layers)
    if corr_mode == "volume":
        # the lookup reads the volume: the named HBM-roofline kernel, measured live in the timed region
        out["roofline"] = lookup_roofline(events, plan.P, tracker.flower.engine.volume_storage)
        if have_conv:
            out["roofline_mfma"] = conv_roofline(conv_events, layers)
    elif have_conv:
        # volume-free correlation: no HBM-bound lookup on the path; the dominant kernel is a matrix-core conv
        # (the volume lookup's HBM roofline is measured in the 'alt_corr' pass below -> 'roofline_lookup')
        out["roofline"] = conv_roofline(conv_events, layers)
    if "roofline" in out and corr_mode != "volume":
        out["roofline"]["events_on_every_nth_frame"] = EVENT_EVERY
    if bool(getattr(plan, "prog_wh", None)) and wh_events:
        out["roofline_weight_head"] = wh_roofline(wh_events, plan.prog_wh[0], plan.P)
    tc_gpu = {}
    if world == 1:
        _, dst, _ = tracker.flower.compute_flow(template, frames[0], mode="TC", do_sigmoid=True)
        tc_gpu[args.precision] = dst.cpu()
        if corr_mode == "otf" and not args.no_alt_corr:
            # the volume-free lookup's cost depends on the spread of the flow inside an 8x8 block (its result never does):
            # the same kernel on this frame's smooth field (live, above) and on a per-pixel scattered field (+-8 px)
            g = torch.Generator(device="cuda").manual_seed(1)
            idx = torch.arange(plan.P, device="cuda")
            grid = torch.stack([idx % plan.wf, idx // plan.wf], 1).float()
            saved = plan.coords.clone()
            res = {}
            for name, field in (("smooth (this sequence)", saved),
                                ("scattered +-3 px", grid + (torch.rand(plan.P, 2, device="cuda", generator=g) * 2 - 1) * 3.0),
                                ("scattered +-8 px", grid + (torch.rand(plan.P, 2, device="cuda", generator=g) * 2 - 1) * 8.0)):
                plan.coords.copy_(field)
                for _ in range(2):
                    ops.run_lookup_otf(plan.lookup)
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for s_, e_ in evs:
                    s_.record()
                    ops.run_lookup_otf(plan.lookup)
                    e_.record()
                torch.cuda.synchronize()
                res[name] = {"avg_launch_us": 1e3 * float(np.mean([s_.elapsed_time(e_) for s_, e_ in evs]))}
            plan.coords.copy_(saved)
            res["in_timed_region_avg_launch_us"] = (1e3 * float(np.mean([s_.elapsed_time(e_) for s_, e_ in events]))
                                                    if events else None)
            out["lookup_otf_by_flow_field"] = res
    tracker = plan = None                 # (frees the main engine's buffers before the extra runs)
    gc.collect()
    torch.cuda.empty_cache()

    K2 = min(K, 40)                        # steps of the extra passes that run "the full K" of a default-sized run

    def side_run(n_steps, check_tracks=False, **kw):
        """Another tracker on the same sequence with the same history; -> timing (+ track identity with the timed run)."""
        trk = make_tracker(kw.pop("precision", args.precision), **kw)
        pl = trk.flower.engine.plan(H, W)
        pl.lookup_events = [] if kw.get("corr") == "volume" else None
        track_all(trk, 0, Wm)
        torch.cuda.synchronize()
        if pl.lookup_events is not None:
            pl.lookup_events.clear()
        t1 = time.perf_counter()
        res = track_all(trk, Wm, n_steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        r = {"frames_per_s": n_steps / dt, "ms_per_step": 1000.0 * dt / n_steps, "steps": n_steps}
        if check_tracks:
            r["tracks_identical_to_timed_run"] = bool(all(np.array_equal(a[0], b_[0]) for a, b_ in zip(res, results[Wm:Wm + n_steps])))
        evs, pl.lookup_events = pl.lookup_events, None
        return r, trk, pl, evs

    def drop(*objs):
        gc.collect()
        torch.cuda.empty_cache()

    if world == 1 and not args.no_alt_precisions:
        # the other arithmetic modes on the same sequence, each its own engine + buffers.  The STRICT fp32 operating
        # point (exact fp32 MFMA products, all-pairs volume: the reference's precision class) runs the full K steps.
        alt = {}
        for prec in ("fp32", "bf16x3", "bf16", "fp16", "f16mx8"):
            if prec == args.precision:
                continue
            r, trk, pl, _ = side_run(K2, precision=prec)
            _, dst, _ = trk.flower.compute_flow(template, frames[0], mode="TC", do_sigmoid=True)
            tc_gpu[prec] = dst.cpu()
            r["correlation"] = trk.flower.engine.corr
            alt[prec] = r
            del trk, pl
            drop()
        out["alt_precisions"] = alt
        if "fp32" in alt:
            out["strict_fp32"] = dict(alt["fp32"], dtype="f32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate)",
                                      note="same workload and step count as the headline, driver-timed in the same run")
    if world == 1 and not args.no_alt_corr and args.precision != "fp32":
        # the other correlation mode, same sequence; in the default (volume-free) mode this is also where the volume
        # lookup -- the named HBM-roofline kernel -- is measured live
        other = "volume" if corr_mode == "otf" else "otf"
        torch.cuda.reset_peak_memory_stats()
        r, trk, pl, evs = side_run(min(K, 8), check_tracks=True, corr=other)
        r["hbm_allocated_peak_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
        out["alt_corr"] = {other: r}
        if other == "volume":
            out["roofline_lookup"] = lookup_roofline(evs, pl.P, trk.flower.engine.volume_storage)
        del trk, pl
        drop()
        if other == "volume" and args.precision != "bf16":
            # the plain-bf16 operating point with its bf16-STORAGE volume (BASELINE config 3's arithmetic; SURVEY 8d:
            # 2096 B per pixel and lookup): the same HBM-roofline kernel on half-size volume elements
            r, trk, pl, evs = side_run(min(K, 8), precision="bf16", corr="volume")
            r["volume_storage"] = trk.flower.engine.volume_storage
            out["alt_corr"]["volume, precision bf16"] = r
            out["roofline_lookup_bf16_storage"] = lookup_roofline(evs, pl.P, trk.flower.engine.volume_storage)
            del trk, pl
            drop()
    if world == 1 and not args.no_alt_corr:
        # the weight head on every pixel (or, with --full-weight-head, on the mask region): full K steps, same tracks
        r, trk, pl, _ = side_run(K2, check_tracks=True, mask_wh=not mask_region)
        out["alt_weight_head"] = {("full" if mask_region else "mask_region"): r}
        del trk, pl
        drop()
        # each flow's launch list replayed as ONE hipGraph (flow config key `graph`; the timed run launches eagerly
        # because its per-launch HIP events cannot live inside a graph): same tracks, short run
        r, trk, pl, _ = side_run(K2, check_tracks=True, graph=True)
        r["graphs_replayed"] = bool(any(g is not None for g in getattr(pl, "_graphs", {}).values()))
        out["alt_graph"] = r
        del trk, pl
        drop()
    if world == 1 and not args.no_ladder:
        # ---- host frames: what a real caller hands track() (WOFT_demo.py:61-78, TRK:113-120) -- one numpy frame per call,
        # crossing PCIe inside the timed region (pinned double-buffered staging + async H2D in woft_amd.tracker._FrameUploader);
        # same sequence, same tracks as the resident-frame headline
        frames_np = [f.cpu().numpy() for f in frames]
        trk = make_tracker(args.precision)
        for i in range(Wm):
            trk.track(frames_np[i % CLIP])
        torch.cuda.synchronize()
        n_hf = max(K2, 20)
        t1 = time.perf_counter()
        res_hf = []
        for i in range(Wm, Wm + n_hf):
            if i > 0 and i % CLIP == 0:
                restart_clip(trk)
            res_hf.append(trk.track(frames_np[i % CLIP]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        out["host_frames"] = {"frames_per_s": n_hf / dt, "ms_per_step": 1000.0 * dt / n_hf, "steps": n_hf,
                              "pcie_bytes_per_frame": int(frames_np[0].nbytes),
                              "tracks_identical_to_timed_run": bool(all(np.array_equal(a[0], b_[0]) for a, b_ in
                                                                        zip(res_hf, results[Wm:Wm + n_hf]))),
                              "note": "numpy frame per track() call: host memcpy into pinned staging + async H2D on the compute "
                                      "stream, inside the timed region; `value` is the resident-frame figure"}
        del trk, frames_np
        drop()
        # ---- the drop-in as a reference user writes it (round-4 review): a tracker config in the REFERENCE's form -- estimator,
        # Sobol subsampler and inlier test defined inline as plain functions, no woft_amd import, no tags
        # (tests/configs/inline_wlsq.py; configs/YAOFT_single_control_repRAFT_sub500_noreliableinl_wLSq.py:14-53) -- (a) with a flow
        # config WITHOUT a `precision` key (built-in default: fp32-emulating bf16x3), (b) with precision = 'bf16x3'.  The
        # tracker recognises the callables by their behaviour (woft_amd.probe) and runs the device solver.  (a') the same with
        # precision = 'fp32': exact products -- what round 4's default for a keyless config was.
        import importlib.util
        spec = importlib.util.spec_from_file_location("inline_wlsq", str(ROOT / "tests" / "configs" / "inline_wlsq.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        ref_form = {}
        for name, prec in (("no_precision_key", None), ("precision_bf16x3", "bf16x3"), ("precision_fp32", "fp32")):
            conf = mod.get_config(precision=prec)
            conf.flow_config.model, conf.flow_config.iters = sd, args.iters
            trk = conf.tracker_class(conf)
            trk.init(template, mask)
            track_all(trk, 0, Wm)
            torch.cuda.synchronize()
            n_rf = max(K2, 20)
            t1 = time.perf_counter()
            res_rf = track_all(trk, Wm, n_rf)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            ref_form[name] = {"frames_per_s": n_rf / dt, "ms_per_step": 1000.0 * dt / n_rf, "steps": n_rf,
                              "precision": trk.flower.precision, "precision_source": trk.flower.precision_source,
                              "solver": trk.solver_decision}
            if trk.flower.precision == args.precision:
                ref_form[name]["tracks_identical_to_timed_run"] = bool(all(np.array_equal(a[0], b_[0]) for a, b_ in
                                                                          zip(res_rf, results[Wm:Wm + n_rf])))
            del trk
            drop()
        out["reference_format_config"] = ref_form
        # ---- like-for-like ladder: the flow operator doing the reference's FULL work (weight head on every pixel, as
        # WeightedRAFT.forward evaluates it) in the fp32-emulating arithmetic of the headline and in the reference's own
        # arithmetic class (exact fp32 MFMA products); same sequence, same step count each
        ladder = {}
        for name, kw in (("bf16x3_full_weight_head", dict(precision="bf16x3", mask_wh=False)),
                         ("fp32_full_weight_head", dict(precision="fp32", mask_wh=False))):
            r, trk, pl, _ = side_run(max(K2, 20), check_tracks=(kw["precision"] == args.precision), **kw)
            r["correlation"] = trk.flower.engine.corr
            ladder[name] = r
            del trk, pl
            drop()
        ladder["note"] = ("headline = bf16x3 + weight head only under the drawn correspondences (tracker-level pruning, identical "
                          "tracks); these two lines remove the pruning, the second also the split-bf16 arithmetic")
        out["reference_work"] = ladder

        # ---- lost frames (TRK:167-207): every 4th frame the re-detection verdict is overruled to 'lost', so the frame
        # also runs the frame t-1 -> t flow (second buffer set: the template stays resident) and its fit.  Per-frame wall
        # time, one host sync per frame: the lost frame, the frame right after it, and the other frames of the same pass
        trk = make_tracker(args.precision)
        every, seen = 4, {"i": 0}
        inner = trk._global_stage

        def overruled(frame, prewarp_H):
            fit = inner(frame, prewarp_H)
            seen["i"] += 1
            if seen["i"] % every == 0:
                fit.success = False
            return fit
        trk._global_stage = overruled
        n_lf = max(K2, 24)
        track_all(trk, 0, Wm)
        seen["i"] = 0
        torch.cuda.synchronize()
        per_frame, metas = [], []
        for i in range(Wm, Wm + n_lf):
            if i > 0 and i % CLIP == 0:
                restart_clip(trk)
            t1 = time.perf_counter()
            _, m_ = trk.track(frames[i % CLIP])
            torch.cuda.synchronize()
            per_frame.append(1000.0 * (time.perf_counter() - t1))
            metas.append(bool(m_.lost))
        lost_ms = [t for t, l in zip(per_frame, metas) if l]
        after_ms = [t for j, (t, l) in enumerate(zip(per_frame, metas)) if not l and j > 0 and metas[j - 1]]
        other_ms = [t for j, (t, l) in enumerate(zip(per_frame, metas)) if not l and not (j > 0 and metas[j - 1])]
        med = lambda v: float(np.median(v)) if v else None
        out["lost_frame"] = {"frames": n_lf, "lost_frames": int(sum(metas)), "forced_every": every,
                             "lost_frame_ms": med(lost_ms), "frame_after_lost_ms": med(after_ms), "normal_frame_ms": med(other_ms),
                             "lost_over_normal": (med(lost_ms) / med(other_ms)) if lost_ms and other_ms else None,
                             "local_stage_weight_head": "deferred to the drawn correspondences" if getattr(trk, "_sparse_weights", False) else "full map",
                             "note": "wall ms per track() with a device sync after every frame (so each is a little above the "
                                     "pipelined ms_per_step of the headline)"}
        # ... and runs of lost frames (a loss usually lasts several frames): in the same pass pattern, frames 2 and 3 of every four are
        # overruled.  From the second frame of a run on, the local flow's source is the previous local flow's target: its features are
        # taken from there instead of encoded again (woft_amd.tracker._local_stage; identical tracks, tested)
        seen["i"], every2 = 0, 4
        trk._global_stage = lambda frame, prewarp_H: _overrule_run(inner, seen, every2, frame, prewarp_H)
        restart_clip(trk)
        first_ms, second_ms, reused = [], [], []
        for i in range(Wm, Wm + 16):
            if i > 0 and i % CLIP == 0:
                restart_clip(trk)
            trk.flower.source_features_reused = False
            t1 = time.perf_counter()
            _, m_ = trk.track(frames[i % CLIP])
            torch.cuda.synchronize()
            ms_ = 1000.0 * (time.perf_counter() - t1)
            if m_.lost and m_.N_lost == 1:
                first_ms.append(ms_)
            elif m_.lost:
                second_ms.append(ms_)
                reused.append(bool(trk.flower.source_features_reused))
        out["lost_frame"]["runs_of_two"] = {"first_lost_frame_ms": med(first_ms), "second_lost_frame_ms": med(second_ms),
                                            "second_over_normal": (med(second_ms) / med(other_ms)) if second_ms and other_ms else None,
                                            "source_features_reused_on_second": bool(reused) and all(reused)}
        del trk
        drop()
        # ---- two sequences on ONE GPU in one process (side figure, never `value`): two trackers, two HIP streams, two host threads
        # (the per-flow result read releases the GIL).  Every layer at 1/8 resolution is ONE round of the chip's 512 workgroup slots,
        # so a single sequence runs its launches in lock-step (common prologue, common store burst); a second stream's launches fill
        # those phases.  What a 16-sequence job on 8 GPUs would get per GPU.
        import threading
        seqs = [(template, frames), make_sequence(H, W, 101, CLIP)]
        streams = [torch.cuda.Stream() for _ in seqs]
        trks2 = []
        for (tpl, frs), st in zip(seqs, streams):
            with torch.cuda.stream(st):
                t2 = make_tracker(args.precision, first=tpl)
                for i in range(Wm):
                    t2.track(frs[i % CLIP])
                st.synchronize()
            trks2.append(t2)
        n2 = K2
        gate = threading.Barrier(len(seqs) + 1)
        errs = []

        def run_sequence(j):
            try:
                with torch.cuda.stream(streams[j]):
                    gate.wait()
                    for i in range(Wm, Wm + n2):
                        if i > 0 and i % CLIP == 0:
                            restart_clip(trks2[j])
                        trks2[j].track(seqs[j][1][i % CLIP])
                    streams[j].synchronize()
            except Exception as ex:          # (reported in the line; the headline does not depend on this pass)
                errs.append(f"{type(ex).__name__}: {ex}")
        threads = [threading.Thread(target=run_sequence, args=(j,)) for j in range(len(seqs))]
        for th in threads:
            th.start()
        torch.cuda.synchronize()
        gate.wait()
        t1 = time.perf_counter()
        for th in threads:
            th.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        out["two_sequences_one_gpu"] = {"sequences": len(seqs), "steps_each": n2, "aggregate_frames_per_s": len(seqs) * n2 / dt,
                                        "ms_per_step_each": 1000.0 * dt / n2, "errors": errs or None,
                                        "note": "two trackers on two HIP streams driven by two host threads of this process; "
                                                "aggregate over both sequences (compare with `value`, one sequence)"}
        del trks2, streams
        drop()
        if K < 200:
            # ---- the driver times --steps frames (0.2 s at 20 steps); the same configuration over >= 2 s, in chunks of 20
            trk = make_tracker(args.precision)
            track_all(trk, 0, Wm)
            torch.cuda.synchronize()
            chunks, t_all = [], time.perf_counter()
            for c in range(10):
                t1 = time.perf_counter()
                track_all(trk, Wm + 20 * c, 20)
                torch.cuda.synchronize()
                chunks.append(20.0 / (time.perf_counter() - t1))
            dt = time.perf_counter() - t_all
            out["steady_state"] = {"steps": 200, "frames_per_s": 200.0 / dt, "ms_per_step": 1000.0 * dt / 200,
                                   "frames_per_s_by_chunk_of_20": {"min": float(np.min(chunks)), "median": float(np.median(chunks)),
                                                                  "max": float(np.max(chunks))}}
            del trk
            drop()
    if world == 1 and not args.no_cpu_baseline:
        if affinity0 is not None:
            # the launch thread was pinned to the GPU's NUMA node for the timed region (woft_amd.dist.bind_to_gpu_node);
            # the CPU baseline gets the cores the process started with
            os.sched_setaffinity(0, affinity0)
        torch.set_num_threads(usable_cores())
        n_cpu = 3
        times, tc = cpu_baseline(sd, template, mask, [frames[i].cpu().numpy() for i in range(n_cpu)], args.iters)
        med = float(np.median(times))
        # quality gate on the first frame: flow EPE of the HIP path(s) against the CPU oracle
        epes = {}
        for prec, dst in tc_gpu.items():
            e = torch.sqrt(((dst - tc[1]).reshape(2, -1) ** 2).sum(0))
            epes[prec] = {"mean_px": float(e.mean()), "max_px": float(e.max())}
        out["cpu_baseline"] = {"value": 1.0 / med, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                               "os_cpu_count": os.cpu_count(), "affinity_cores": usable_cores(), "seconds_per_frame": [round(t, 2) for t in times],
                               "sample": f"{n_cpu} tracked frames at {H}x{W}, {args.iters} iters (oracle/tracker_ref.py, torch-CPU "
                                         f"fp32 restatement of the reference path), median {med:.1f} s per frame"}
        out["flow_epe_vs_cpu_oracle"] = epes[args.precision]
        out["flow_epe_vs_cpu_oracle_by_precision"] = epes
        # parity gate (SURVEY 8d budgets, flow EPE in px against the CPU oracle: (mean, max)): a run outside the budget of
        # its precision is a FAILED run, whatever its frames/s
        scale = 1.0 if args.iters <= 12 else 3.0          # (8d: bf16 0.05 px @ 12 it, 0.15 px @ 32 it)
        bad = {}
        for prec, e in epes.items():
            b_mean, b_max = EPE_BUDGET[prec]
            e["budget_px"] = {"mean": b_mean * scale, "max": b_max * scale}
            e["within_budget"] = bool(e["mean_px"] <= b_mean * scale and e["max_px"] <= b_max * scale)   # (NaN -> False)
            if not e["within_budget"]:
                bad[prec] = e
        out["epe_gate"] = {"passed": not bad, "failed_precisions": sorted(bad)}
        gate_failed = bool(bad)
    else:
        out["epe_gate"] = {"passed": None, "note": "skipped: no CPU-oracle pass in this run (--no-cpu-baseline or N > 1)"}
        gate_failed = False
    # the figures a reader of the driver's `parsed` record needs beside `value` (it keeps `config`, not the extra keys)
    cfgd = out["config"]
    g = lambda d, *ks: (g(d.get(ks[0], {}), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    r1 = lambda v: None if v is None else round(float(v), 2)
    cfgd["fps_strict_fp32_full_head"] = r1(g(out, "reference_work", "fp32_full_weight_head", "frames_per_s"))
    cfgd["fps_bf16x3_full_head"] = r1(g(out, "reference_work", "bf16x3_full_weight_head", "frames_per_s"))
    cfgd["fps_strict_fp32"] = r1(g(out, "strict_fp32", "frames_per_s"))
    cfgd["fps_host_frames"] = r1(g(out, "host_frames", "frames_per_s"))
    cfgd["fps_reference_form_config_unmodified"] = r1(g(out, "reference_format_config", "no_precision_key", "frames_per_s"))
    cfgd["fps_reference_form_config_bf16x3"] = r1(g(out, "reference_format_config", "precision_bf16x3", "frames_per_s"))
    cfgd["fps_reference_form_config_fp32"] = r1(g(out, "reference_format_config", "precision_fp32", "frames_per_s"))
    cfgd["lost_run_second_over_normal"] = g(out, "lost_frame", "runs_of_two", "second_over_normal")
    cfgd["fps_two_sequences_one_gpu"] = r1(g(out, "two_sequences_one_gpu", "aggregate_frames_per_s"))
    cfgd["fps_f16mx8"] = r1(g(out, "alt_precisions", "f16mx8", "frames_per_s"))     # (opt-in: 2 matrix-pipe passes per product on the 3x3 layers)
    cfgd["epe_mean_px"] = g(out, "flow_epe_vs_cpu_oracle", "mean_px")
    cfgd["epe_gate_passed"] = out["epe_gate"]["passed"]
    print(json.dumps(out), flush=True)
    if gate_failed:
        print(f"bench.py: flow EPE outside the SURVEY 8d budget: {json.dumps(bad)}", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    _clean = False
    try:
        main()
        _clean = True
    except SystemExit as ex:             # (rank 0's EPE-gate exit: the other ranks are already waiting at the last barrier)
        _clean = True
        raise
    finally:
        from woft_amd import dist as _wdist
        _wdist.finalize(clean=_clean)
