#!/usr/bin/env python3
"""bench.py -- Mpix/s of SIFT extraction on synthetic 1920x1080 grayscale (BASELINE.json config 2 / 4).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Started WITHOUT a torch.distributed environment and with --gpus N > 1, this script starts its
N rank processes itself (before it has imported torch or touched HIP) and relays rank 0's line.

A "step" is one pass of the extraction hot path over one batch of --batch synthetic images per GPU (inputs already
resident in HBM; --contexts extraction contexts, each with its own HIP stream, work through the batch --launch-batch
images at a time -- popsift_hip_submit_batch: every kernel launched once for those images; results left device
resident like the reference's FeaturesDev).  Images are independent, so
ranks never talk on the data path (weak scaling, no RCCL); torch.distributed is used for the barriers around the
timed region and the MAX over ranks only.

Rank 0 prints ONE JSON line with the driver contract fields plus
  roofline      -- the WHOLE pipeline of one image against the HBM roofline (B_alg / T_dev, T_dev from
                   HIP events on the context's stream), with a `kernels` list: per stage its device time (HIP events
                   between the launches, C-ABI profile mode 2), share, bound, achieved / peak / frac and the HBM
                   traffic the rocprofv3 counter passes measured for it (profiles/r04_kernel_counters.json; quoted only
                   while the kernel sources are the ones that were counted, "counters_stale" otherwise)
  cpu_baseline  -- the CPU oracle (kind "port") on a bounded sample, N=1 only
  sparse_image  -- the same pipeline on a keypoint-sparse image (about 2 features per 1000 pixels), where the pyramid
                   -- the part `north_star` calls bandwidth-bound -- carries the time
  legs_s        -- wall time of every leg of this run (the timed region is `timed`)
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

W, H = 1920, 1080
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_GBPS = 6290.0  # ... and the read ceiling measured on the device
VALU_PEAK_GINST = 1228.8    # 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md)
# What a kernel of nothing but independent v_fma_f32, 8 waves per SIMD on every CU, sustains for 20 .. 100 ms on this device
# (tools/ubench/clock_under_load.hip, profiles/r04_clock_under_load.txt: 1.010e12 wave64 instructions per second = 129
# TFLOP/s, 0.82 of the nominal peak: the clock the package power allows)
VALU_SUSTAINED_GINST = 1010.0
# Issue slots per vector instruction of k_descriptor's sample loop (tools/isa_count.py: 97 instructions = 120.4 full-rate
# slots; compares / conversions / min / max count 1.76, transcendentals 3.46: tools/ubench/valu_rate.hip)
DESC_SLOTS_PER_INSTR = 120.4 / 97.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per step per GPU (one step = one batch)")
    ap.add_argument("--contexts", type=int, default=3, help="extraction contexts (streams) per GPU")
    ap.add_argument("--launch-batch", type=int, default=16,
                    help="images a context extracts per submit (popsift_hip_submit_batch: every kernel launched once for all of them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="timed loop only: no roofline / host-to-host / CPU legs (A/B runs)")
    ap.add_argument("--only-roofline", action="store_true",
                    help="run only the single-context passes (the command the profiles/ *_single_image* files were taken with)")
    ap.add_argument("--cpu-images", type=int, default=6, help="images timed for the CPU baseline")
    ap.add_argument("--debug", default="", help="tuning runs: popsift_hip_debug_set switches, what:value,...")
    ap.add_argument("--size", default=None, help="tuning runs (--quick): WxH of the synthetic images instead of 1920x1080 "
                                                 "(3840x2160: BASELINE.json config 3)")
    ap.add_argument("--threshold", type=float, default=None,
                    help="tuning runs (--quick): Config threshold of the timed loop (0.17: the keypoint-sparse regime)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a torch.distributed environment: be the launcher.  Nothing here imports torch or touches the
    GPU, so no process that has initialised HIP is ever replaced or forked."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(args.gpus))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def kernel_counters():
    """Per-stage HBM traffic and instruction counts from the rocprofv3 counter passes of this configuration
    (tools/collect_profiles.sh -> profiles/r04_kernel_counters.json; FETCH_SIZE x2 + WRITE_SIZE, separate passes).
    Counters cannot be collected inside the timed process, so the committed measurement is quoted -- as long as it is a
    measurement of THESE kernels: the summary carries the hash of the kernel sources it was taken from
    (popsift_amd/srchash.py); when the tree's differs the counters are marked stale and nothing is derived from them."""
    try:
        with open(os.path.join(HERE, "profiles", "r04_kernel_counters.json")) as f:
            ctr = json.load(f)
    except Exception:
        return {}, True
    from popsift_amd.srchash import kernel_source_hash
    return ctr, ctr.get("_kernel_source_hash") != kernel_source_hash()


class Workers:
    """One persistent thread per extraction context.  run(k) works through k steps -- k batches of len(ptrs) images, one
    after the other without a pause between them: every context takes the next `launch_batch` images of the stream
    (submit, wait, next), so no context idles at a step boundary while another finishes its share.  The call returns
    when the last image of the last step is done."""

    def __init__(self, ctxs, ptrs, launch_batch=1, size=None):
        self.ctxs, self.ptrs, self.lb = ctxs, ptrs, max(1, launch_batch)
        self.w, self.h = size if size else (W, H)
        self.n = len(ctxs)
        self.go = threading.Barrier(self.n + 1)
        self.done = threading.Barrier(self.n + 1)
        self.stop = False
        self.lock = threading.Lock()
        self.next = self.end = 0
        self.feats = [0] * self.n
        self.descs = [0] * self.n
        self.threads = [threading.Thread(target=self._loop, args=(i,), daemon=True) for i in range(self.n)]
        for t in self.threads:
            t.start()

    def _take(self):
        """the next images of the stream: [first, last) or None"""
        with self.lock:
            if self.next >= self.end:
                return None
            # a launch does not straddle two steps (a step is one batch of the workload)
            step_end = (self.next // len(self.ptrs) + 1) * len(self.ptrs)
            a, b = self.next, min(self.next + self.lb, step_end, self.end)
            self.next = b
            return a, b

    def _loop(self, i):
        ctx = self.ctxs[i]
        n = len(self.ptrs)
        while True:
            self.go.wait()
            if self.stop:
                return
            f = d = 0
            while True:
                t = self._take()
                if t is None:
                    break
                mine = [self.ptrs[k % n] for k in range(*t)]
                if self.lb == 1:
                    ctx.submit_dev(mine[0], self.w, self.h, self.w)
                    nf, nd = ctx.wait()
                    f += nf
                    d += nd
                else:
                    ctx.submit_batch_dev(mine, self.w, self.h, self.w)
                    for nf, nd in ctx.wait_batch():
                        f += nf
                        d += nd
            self.feats[i], self.descs[i] = f, d
            self.done.wait()

    def run(self, steps):
        self.next, self.end = 0, steps * len(self.ptrs)
        self.go.wait()
        self.done.wait()

    def close(self):
        self.stop = True
        self.go.wait()
        for t in self.threads:
            t.join()


def b_alg(rep, w, h, in_bytes=1):
    """input read once, every Gaussian plane written and read once, every DoG plane written and read
    once, outputs written once (88 B per pyramid pixel).  The DoG planes are not stored by this build (their consumers
    subtract Gaussian planes), which moves fewer bytes; the survey's formula is kept as the yardstick."""
    L = 6
    return w * h * in_bytes + 4.0 * rep.pyramid_pixels * (2 * L + 2 * (L - 1)) + 72.0 * rep.ext_total + 512.0 * rep.ori_total


def b_alg_this_build(rep, w, h, in_bytes=1):
    """The same for this build's own layout: no DoG plane is written or read; every Gaussian plane is written once and read
    twice (by the next level and by detection; the other consumers gather) -- 12 B x L = 72 B per pyramid pixel minus the
    first plane's missing producer read: 68 B -- plus input and outputs."""
    L = 6
    return w * h * in_bytes + 4.0 * rep.pyramid_pixels * (3 * L - 1) + 72.0 * rep.ext_total + 512.0 * rep.ori_total


def single_image(ctx, ptr, hip, n=5):
    """T_dev (HIP events, first to last kernel, median of n) and the stage times (profile mode 2) of one image."""
    import numpy as np
    lat = []
    for _ in range(n):
        ctx.submit_dev(ptr, W, H, W)
        ctx.wait()
        lat.append(ctx.report().ms_device)
    ctx.set_profile(2)
    st = []
    for _ in range(n):
        ctx.submit_dev(ptr, W, H, W)
        ctx.wait()
        st.append(list(ctx.report().ms_stage)[:len(hip.STAGES)])
    ctx.set_profile(0)
    rep = ctx.report()
    return float(np.median(lat)), [float(x) for x in np.median(np.array(st), 0)], rep


def other_config(hip, torch, np, synth, device, w, h, seed, params_kw, contexts, launch_batch, images_per_step, distinct, steps=3):
    """Reported extra (never `value`): another BASELINE.json configuration -- one image alone on the device (T_dev by HIP
    events, median of 5, and the pipeline's algorithmic bytes over it) and the same in-flight arrangement as the timed loop
    (`contexts` contexts x `launch_batch` images per launch)."""
    imgs = [synth(seed + k, w, h) for k in range(distinct)]
    dev = [torch.from_numpy(im).cuda(device) for im in imgs]
    ptrs = [dev[i % distinct].data_ptr() for i in range(images_per_step)]
    ctxs = [hip.Context(hip.default_params(**params_kw), device=device) for _ in range(contexts)]
    lat = []
    for _ in range(5):
        ctxs[0].submit_dev(ptrs[0], w, h, w)
        ctxs[0].wait()
        lat.append(ctxs[0].report().ms_device)
    rep = ctxs[0].report()
    ms_dev = float(np.median(lat))
    bytes_alg = b_alg(rep, w, h)
    wk = Workers(ctxs, ptrs, launch_batch, size=(w, h))
    wk.run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wk.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    wk.close()
    for c in ctxs:
        c.close()
    n = steps * images_per_step
    return {"size": "%dx%d" % (w, h), "params": params_kw, "octaves": rep.num_octaves, "base_plane": "%dx%d" % (rep.base_w, rep.base_h),
            "features": rep.ext_total, "descriptors": rep.ori_total,
            "features_per_1000_px": round(rep.ext_total / (w * h / 1000.0), 2),
            "single_image_ms_device": round(ms_dev, 4),
            "single_image_pipeline_frac_of_8TBps": round(bytes_alg / (ms_dev * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "in_flight_contexts": contexts, "images_per_launch": launch_batch,
            "in_flight_mpix_s": round(n * w * h / 1e6 / dt, 1), "in_flight_ms_per_image": round(dt / n * 1e3, 4),
            "in_flight_pipeline_frac_of_8TBps": round(bytes_alg * n / dt / 1e9 / HBM_PEAK_GBPS, 4)}


def main():
    global W, H
    args = parse()
    if args.size:
        if not args.quick:
            raise SystemExit("bench.py: --size is for --quick tuning runs; the bench line is BASELINE.json's 1920x1080")
        W, H = (int(v) for v in args.size.lower().split("x"))
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    legs = {}
    t_leg = time.perf_counter()

    def leg(name):
        nonlocal t_leg
        now = time.perf_counter()
        legs[name] = round(legs.get(name, 0.0) + now - t_leg, 3)
        t_leg = now

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: WORLD_SIZE=%d but --gpus %d: reporting %d ranks" % (world, args.gpus, world), file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the extraction path has no CPU fallback)")
    # Rehearsal knobs (1-GPU box only): BENCH_FORCE_DEVICE=0 puts every rank on one card and
    # BENCH_BACKEND=gloo swaps RCCL for gloo, to exercise the N>1 control path without N GPUs.
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants GPU %d but only %d are visible (one process per GPU; set "
                         "BENCH_FORCE_DEVICE=0 BENCH_BACKEND=gloo to rehearse the control path on one card)"
                         % (rank, local_rank, torch.cuda.device_count()))
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from popsift_amd import _capi as hip
    from popsift_amd.shard import reduce_stats, shard_indices
    from popsift_amd.synth import synth
    leg("import_init")

    B = 1 if args.only_roofline else args.batch
    # BASELINE.json config 4: a batch of seeds 100.. dealt to the ranks (popsift_amd/shard.py: image i -> rank i mod N);
    # config 2's own image (seed 2) is image 0 of rank 0.  A rank synthesises min(B, 16) distinct images (0.5 s of host
    # time each) and cycles through them.
    U = min(B, 16)
    mine = shard_indices(world * U, rank, world)
    seeds = [2 if (rank == 0 and k == 0) else 100 + i for k, i in enumerate(mine)]
    host_imgs = [synth(s, W, H) for s in seeds]
    leg("synth")

    dev_imgs = [torch.from_numpy(im).cuda(local_rank) for im in host_imgs]  # inputs resident in HBM
    ptrs = [dev_imgs[i % U].data_ptr() for i in range(B)]
    C = max(1, min(args.contexts, B))
    pkw = {} if args.threshold is None else {"threshold": args.threshold}
    ctxs = [hip.Context(hip.default_params(**pkw), device=local_rank) for _ in range(C)]
    if args.debug:      # tuning runs (tools/): popsift_hip_debug_set switches "what:value,..."
        for item in args.debug.split(","):
            what, value = item.split(":")
            for c in ctxs:
                c.debug_set(int(what), int(value))
    workers = Workers(ctxs, ptrs, args.launch_batch)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if args.only_roofline:
        args.steps, args.warmup = 1, 0
    leg("setup")
    if args.warmup:
        workers.run(args.warmup)
    barrier()
    leg("warmup")
    t0 = time.perf_counter()
    workers.run(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    leg("timed")
    workers.close()

    # MAX over ranks of the time, SUM of the counters (popsift_amd/shard.py)
    elapsed, (feats_step, descs_step) = reduce_stats(dist if world > 1 else None, elapsed,
                                                     (sum(workers.feats), sum(workers.descs)),
                                                     device="cuda" if backend == "nccl" else "cpu")
    feats_step, descs_step = feats_step / max(args.steps, 1), descs_step / max(args.steps, 1)   # run() counts all steps
    images = args.steps * B * world
    mpix = images * W * H / 1e6
    value = mpix / elapsed

    extra = {}
    roofline = cpu = None
    if rank == 0 and not args.quick:
        c0 = ctxs[0]
        # ---- one image on an otherwise idle GPU: T_dev, stage times, pipeline roofline ------------------------------
        ms_dev, stage_ms, rep = single_image(c0, ptrs[0], hip)
        bytes_alg = b_alg(rep, W, H)
        bytes_own = b_alg_this_build(rep, W, H)
        achieved = bytes_alg / (ms_dev * 1e-3) / 1e9
        ctr, stale = kernel_counters()
        kernels = []
        ssum = sum(stage_ms) or 1.0
        for name, ms in zip(hip.STAGES, stage_ms):
            k = {"stage": name, "ms": round(ms, 4), "share": round(ms / ssum, 3)}
            c = {} if stale else ctr.get(name, {})
            k["traffic"] = c.get("hbm_bytes")
            if name in ("orientation", "descriptor", "scan"):
                # VALU-issue bound: wave64 vector instructions per second against 1 per 2 cycles per SIMD
                k["bound"] = "valu"
                if c.get("valu_insts") and ms > 0:
                    k["achieved"] = round(c["valu_insts"] / (ms * 1e-3) / 1e9, 1)
                    k["peak"], k["unit"] = VALU_PEAK_GINST, "G wave-instr/s"
                    k["frac"] = round(k["achieved"] / VALU_PEAK_GINST, 4)
                    # against the issue rate the device sustains, in full-rate issue slots where the mix is known
                    slots = DESC_SLOTS_PER_INSTR if name == "descriptor" else 1.0
                    k["sustained_peak"], k["slots_per_instr"] = VALU_SUSTAINED_GINST, round(slots, 3)
                    k["frac_of_sustained"] = round(k["achieved"] * slots / VALU_SUSTAINED_GINST, 4)
            else:
                k["bound"] = "hbm"
                alg = {"pyramid": W * H + 4.0 * rep.pyramid_pixels * 2 * 6,           # every plane written + read once
                       "detect": 4.0 * rep.pyramid_pixels * 6,                          # six Gaussian planes read once
                       "refine": None}[name]
                if alg and ms > 0:
                    k["alg_bytes"] = alg
                    k["achieved"] = round(alg / (ms * 1e-3) / 1e9, 1)
                    k["peak"], k["unit"] = HBM_PEAK_GBPS, "GB/s"
                    k["frac"] = round(k["achieved"] / HBM_PEAK_GBPS, 4)
            kernels.append(k)
        traffic = sum(k["traffic"] for k in kernels if k.get("traffic")) or None
        ms_img = elapsed / (args.steps * B) * 1e3
        valu_img = None if stale else (sum(v.get("valu_insts", 0.0) for n, v in ctr.items() if isinstance(v, dict)) or None)
        roofline = {
            "kernel": "whole pipeline, one 1920x1080 image (pyramid -> detect -> refine -> orientation -> descriptors)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "frac_of_measured_ceiling": round(achieved / HBM_MEASURED_GBPS, 4), "measured_ceiling": HBM_MEASURED_GBPS,
            "alg_bytes_per_image": bytes_alg, "ms_device": round(ms_dev, 4),
            # the same against this build's own compulsory traffic (no DoG planes): 68 instead of 88 B per pyramid pixel
            "this_build": {"alg_bytes_per_image": bytes_own, "achieved": round(bytes_own / (ms_dev * 1e-3) / 1e9, 1),
                           "frac": round(bytes_own / (ms_dev * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                           "traffic_over_alg": round(traffic / bytes_own, 3) if traffic else None},
            "steady_state": {"ms_per_image": round(ms_img, 4),
                             "achieved": round(bytes_alg / (ms_img * 1e-3) / 1e9, 1),
                             "frac": round(bytes_alg / (ms_img * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "frac_of_measured_ceiling": round(bytes_alg / (ms_img * 1e-3) / 1e9 / HBM_MEASURED_GBPS, 4),
                             # vector instructions of all kernels of an image (counter passes) over the timed rate
                             "valu_issue_frac": round(valu_img / (ms_img * 1e-3) / 1e9 / VALU_PEAK_GINST, 4) if valu_img else None,
                             "valu_issue_frac_of_sustained": round(valu_img / (ms_img * 1e-3) / 1e9 / VALU_SUSTAINED_GINST, 4) if valu_img else None,
                             "hbm_traffic_frac": round(traffic / (ms_img * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None},
            "counters": "profiles/r04_kernel_counters.json" if not stale else None, "counters_stale": stale,
            "dominant": max(kernels, key=lambda k: k["ms"])["stage"], "kernels": kernels,
        }
        extra["single_image"] = {"ms_device": round(ms_dev, 4), "features": rep.ext_total, "descriptors": rep.ori_total}
        leg("single_image")
        # ---- the blur level launches alone (profile mode 1: every launch timed) -----------------------------------
        c0.set_profile(1)
        big_ms = big_bytes = all_ms = all_bytes = 0.0
        big_n = all_n = 0
        for _ in range(5):
            c0.submit_dev(ptrs[0], W, H, W)
            c0.wait()
            r = c0.report()
            big_ms += r.ms_big
            big_bytes += r.big_alg_bytes
            big_n += r.big_launches
            all_ms += r.ms_blur
            all_bytes += r.blur_alg_bytes
            all_n += r.blur_launches
        c0.set_profile(0)
        if big_n:
            roofline["blur_level_launch"] = {
                "kernel": "k_blur_tile<HALO,0,64> (fused H+V Gaussian level, octave 0)", "launches": big_n,
                "avg_launch_us": round(big_ms * 1e3 / big_n, 2), "alg_bytes_per_launch": round(big_bytes / big_n, 1),
                "achieved": round(big_bytes / (big_ms * 1e-3) / 1e9, 1),
                "frac": round(big_bytes / (big_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "all_blur_launches": {"launches": all_n, "avg_launch_us": round(all_ms * 1e3 / max(all_n, 1), 2),
                                      "achieved": round(all_bytes / (all_ms * 1e-3) / 1e9, 1) if all_ms > 0 else 0.0}}
        # ... and in a launch of the timed loop's size (the strip-march kernels, blur_march.hip): per PLANE
        LBp = max(1, min(args.launch_batch, len(ptrs)))
        if big_n and LBp > 1:
            c0.set_profile(1)
            b_ms = b_bytes = 0.0
            b_n = 0
            for _ in range(3):
                c0.submit_batch_dev([ptrs[k % len(ptrs)] for k in range(LBp)], W, H, W)
                c0.wait_batch()
                r = c0.report()
                b_ms += r.ms_big
                b_bytes += r.big_alg_bytes
                b_n += r.big_launches
            c0.set_profile(0)
            if b_n and b_ms > 0:
                roofline["blur_level_launch"]["in_a_batch"] = {
                    "kernel": "k_blur_march<HALO> (the same level, %d images per launch)" % LBp, "launches": b_n,
                    "avg_us_per_plane": round(b_ms * 1e3 / b_n / LBp, 2), "achieved": round(b_bytes / (b_ms * 1e-3) / 1e9, 1),
                    "frac": round(b_bytes / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        leg("blur_profile")
        # ---- a keypoint-sparse image: the pyramid-bound regime ------------------------------------------------------
        if not args.only_roofline:
            sp = hip.Context(hip.default_params(threshold=0.17), device=local_rank)
            lat = []
            for _ in range(5):
                sp.submit_dev(ptrs[0], W, H, W)
                sp.wait()
                lat.append(sp.report().ms_device)
            ms_sp = float(np.median(lat))
            rs = sp.report()
            bs = b_alg(rs, W, H)
            extra["sparse_image"] = {"threshold": 0.17, "features": rs.ext_total, "descriptors": rs.ori_total,
                                     "features_per_1000_px": round(rs.ext_total / (W * H / 1000.0), 2),
                                     "ms_device": round(ms_sp, 4), "pipeline_alg_GBps": round(bs / (ms_sp * 1e-3) / 1e9, 1),
                                     "pipeline_frac_of_8TBps": round(bs / (ms_sp * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
            sp.close()
            # ... and its throughput in the configuration of the timed loop (reported extra; pyramid + detection set the
            # pace here): 16 x 1 / 3 x 8 contexts x images per launch: 8.2 / 9.0 Gpix/s (tools/batch_sweep.sh)
            sp_ctxs = [hip.Context(hip.default_params(threshold=0.17), device=local_rank) for _ in range(C)]
            spw = Workers(sp_ctxs, ptrs, args.launch_batch)
            n_sp = 8
            spw.run(3)   # the single-image legs before this one leave the device idle: let the clocks come back
            torch.cuda.synchronize()
            t_sp = time.perf_counter()
            spw.run(n_sp)
            torch.cuda.synchronize()
            dt_sp = time.perf_counter() - t_sp
            spw.close()
            for c in sp_ctxs:
                c.close()
            extra["sparse_image"]["in_flight_contexts"] = len(sp_ctxs)
            extra["sparse_image"]["images_per_launch"] = args.launch_batch
            extra["sparse_image"]["in_flight_mpix_s"] = round(n_sp * len(ptrs) * W * H / 1e6 / dt_sp, 1)
            extra["sparse_image"]["in_flight_ms_per_image"] = round(dt_sp / (n_sp * len(ptrs)) * 1e3, 4)
            extra["sparse_image"]["in_flight_pipeline_frac_of_8TBps"] = round(
                bs * n_sp * len(ptrs) / dt_sp / 1e9 / HBM_PEAK_GBPS, 4)
            leg("sparse_image")
        # ---- BASELINE.json configs 3 and 1 in the arrangement of the timed loop (reported extras) -----------------------
        if not args.only_roofline:
            # config 3: 3840x2160, default Config (2x upscale: base plane 7680x4320, 10 octaves); 1.1 GB of planes per image
            extra["config3_4k"] = other_config(hip, torch, np, synth, local_rank, 3840, 2160, 3, {}, C, min(args.launch_batch, 4),
                                               3 * min(args.launch_batch, 4), 2)
            # config 1: 640x480, three octaves, VLFeat mode
            extra["config1_vga"] = other_config(hip, torch, np, synth, local_rank, 640, 480, 1, {"octaves": 3, "sift_mode": 2}, C,
                                                args.launch_batch, 256, 8)
            leg("configs_1_3")
        # ---- PCIe-inclusive end-to-end rate (host image in, host features out), one context -----------------------
        t1 = time.perf_counter()
        n_e2e = 0 if args.only_roofline else 8
        for k in range(n_e2e):
            c0.submit(host_imgs[0])
            c0.fetch()
        if n_e2e:
            extra["host_to_host_single_ctx_mpix_s"] = round(n_e2e * W * H / 1e6 / (time.perf_counter() - t1), 1)
        leg("h2h_single_ctx")

    for c in ctxs:
        c.close()
    del dev_imgs
    torch.cuda.empty_cache()
    if world > 1:
        dist.barrier()          # every rank has released its contexts: the C++ leg below gets the GPUs to itself

    if rank == 0 and not args.quick and not args.only_roofline:
        # the drop-in C++ API (PopSift::enqueue ... SiftJob::get, pinned result pool) over EVERY GPU of the job:
        # a child process with one worker pool over POPSIFT_DEVICES; reported, never `value`
        exe = os.path.join(HERE, "popsift_amd", "popsift-bench")
        if os.path.exists(exe):
            devs = ",".join(str(d) for d in range(world)) if os.environ.get("BENCH_FORCE_DEVICE") is None else \
                ",".join([os.environ["BENCH_FORCE_DEVICE"]] * world)
            try:
                import tempfile
                tmp = tempfile.mkdtemp(prefix="popsift_bench_")
                pgms = []
                for k, im in enumerate(host_imgs[:8]):   # the synth.py images of this workload, as PGM files
                    pgms.append(os.path.join(tmp, "img%d.pgm" % k))
                    with open(pgms[-1], "wb") as f:
                        f.write(b"P5\n%d %d\n255\n" % (W, H))
                        f.write(im.tobytes())
                def cpp_leg(per_dev, more=(), jobs_per_submit=1):
                    inflight = (4 * per_dev * jobs_per_submit + (8 if jobs_per_submit > 1 else 0)) * world
                    r = subprocess.run([exe, "--images", str(64 * world), "--inflight", str(inflight),
                                        "--callers", str(max(2, world)), "--pgm", ",".join(pgms)] + list(more),
                                       capture_output=True, text=True, timeout=180,
                                       env=dict(os.environ, POPSIFT_CONTEXTS_PER_DEVICE=str(per_dev), POPSIFT_DEVICES=devs,
                                                POPSIFT_BATCH=str(jobs_per_submit),
                                                # every job in flight holds ~82 MB of pinned result blocks: let the pool keep them
                                                POPSIFT_PINNED_CACHE_MB=str((inflight + 2 * per_dev * world + 4) * 100)))
                    out = json.loads(r.stdout.strip().splitlines()[-1])
                    out["devices"], out["jobs_per_submit"] = devs, jobs_per_submit
                    return out
                # This process has just released ~50 GB of context buffers, and for the next few seconds the kernels of ANOTHER
                # process run slow (the driver clears released device memory on the GPU): whichever child started first read
                # low in EVERY pass of popsift-bench (warmup_passes_mpix_s) -- the dense leg 1.08-1.10 Gpix/s where the same
                # command repeated later reads 2.03-2.08, the sparse leg 8.1 against 9.1 -- while the PCIe link, measured
                # from here, delivered its 57 GB/s throughout; after a pause of 4 s the first child reads what the later
                # ones do.  The legs measure the C++ API, not this process's tear-down: wait.
                time.sleep(5.0)
                # the keypoint-sparse regime (threshold 0.17, ~2 features per 1000 px: results of a few MB per image, so the
                # PCIe link is not the limit; a worker takes up to 8 queued jobs per submit, POPSIFT_BATCH)
                extra["host_to_host_cpp_api_sparse"] = cpp_leg(3, ["--threshold", "0.17", "--images", str(512 * world)], 8)
                extra["host_to_host_cpp_api"] = cpp_leg(4)
                # one context per GPU: its download of image i runs under the kernels of image i+1 (fetch_begin / fetch_end)
                extra["host_to_host_cpp_api_one_context"] = cpp_leg(1)
            except Exception as e:  # a reported extra: never fail the bench line over it
                extra["host_to_host_cpp_api"] = {"error": str(e)[:200]}
        leg("cpp_api")
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O  # checker / reported baseline only
            # a 1-GPU box grants a 16-core CPU share; more OpenMP threads than that only oversubscribe
            cores = min(os.cpu_count() or 1, 16)
            orc = O.Oracle(O.default_params(), threads=cores)
            orc.run(host_imgs[0])  # warm-up (allocations)
            t1 = time.perf_counter()
            for k in range(args.cpu_images):
                orc.run(host_imgs[k % len(host_imgs)])
            dt = time.perf_counter() - t1
            cpu = {"value": round(args.cpu_images * W * H / 1e6 / dt, 3), "unit": "Mpix/s", "cores": cores,
                   "kind": "port",
                   "sample": "%d x 1920x1080 synthetic images, full pipeline, CPU restatement of PopSift "
                             "(oracle/, OpenMP, %d threads)" % (args.cpu_images, cores)}
            leg("cpu_baseline")

    if rank == 0:
        if args.quick:
            out = {"value": round(value, 2), "unit": "Mpix/s", "n_gpus": world,
                   "ms_per_step": round(elapsed / args.steps * 1e3, 4), "steps": args.steps, "quick": True}
        else:
            out = {
                "metric": "Mpix/s SIFT extract on 1920x1080 (keypoints+descriptors)",
                "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                "data": "synthetic",
                "config": {"workload": "1920x1080 u8 grayscale, default popsift::Config (2x upscale, 9 octaves, "
                                       "3 levels, PopSift mode, loop descriptor, RootSift)",
                           "images_per_step_per_gpu": B, "distinct_images_per_gpu": U, "contexts_per_gpu": C,
                           "images_per_launch": args.launch_batch,
                           "features_per_1000_px": round(feats_step / max(B * world, 1) / (W * H / 1000.0), 2),
                           "results": "device resident (features + descriptors)"},
                "features_per_s": round(feats_step * args.steps / elapsed, 1),
                "descriptors_per_s": round(descs_step * args.steps / elapsed, 1),
                "roofline": roofline, "cpu_baseline": cpu,
            }
            out.update(extra)
            out["legs_s"] = legs
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
