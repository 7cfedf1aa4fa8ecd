#!/usr/bin/env python3
"""bench.py -- Mpix/s of SIFT extraction on synthetic 1920x1080 grayscale (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A "step" is one pass of the extraction hot path over one
batch of --batch synthetic images per GPU (inputs already resident in HBM;
--contexts extraction contexts, each with its own HIP stream, work through the
batch with one image in flight per context; results left device resident like
the reference's FeaturesDev).  Images are independent, so ranks
never talk on the data path (weak scaling, no RCCL); torch.distributed is used
for the barriers around the timed region and the MAX over ranks only.

Rank 0 prints ONE JSON line with the driver contract fields plus
  roofline     -- the blur-level kernel (k_blur_tile), timed live with HIP events
                  on the kernel's own stream (C-ABI profile mode)
  cpu_baseline -- the CPU oracle (kind "port") on a bounded sample, N=1 only
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

W, H = 1920, 1080
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per step per GPU (one step = one batch)")
    ap.add_argument("--contexts", type=int, default=16, help="extraction contexts (images in flight) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="timed loop only: no roofline / host-to-host / CPU legs (A/B runs)")
    ap.add_argument("--only-roofline", action="store_true",
                    help="run only the single-context roofline pass (the command profiles/ *_roofline_pass* was taken with)")
    ap.add_argument("--cpu-images", type=int, default=6, help="images timed for the CPU baseline")
    return ap.parse_args()


def blur_traffic():
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of this configuration
    (FETCH_SIZE x2 + WRITE_SIZE, separate passes; tools/collect_profiles.sh -> profiles/blur_traffic.json).
    PMC collection cannot run inside the timed process, so the committed measurement is quoted."""
    try:
        with open(os.path.join(HERE, "profiles", "blur_traffic.json")) as f:
            return round(json.load(f)["traffic_bytes_per_launch"], 1)
    except Exception:
        return None


class Workers:
    """One persistent thread per extraction context.  A step hands every context its share of the batch
    (images i, i + C, i + 2C, ... of the step), which the context extracts back to back; the step ends when
    every context has finished its last image."""

    def __init__(self, ctxs, ptrs):
        self.ctxs, self.ptrs = ctxs, ptrs
        self.n = len(ctxs)
        self.go = threading.Barrier(self.n + 1)
        self.done = threading.Barrier(self.n + 1)
        self.stop = False
        self.feats = [0] * self.n
        self.descs = [0] * self.n
        self.threads = [threading.Thread(target=self._loop, args=(i,), daemon=True) for i in range(self.n)]
        for t in self.threads:
            t.start()

    def _loop(self, i):
        ctx = self.ctxs[i]
        while True:
            self.go.wait()
            if self.stop:
                return
            f = d = 0
            for p in self.ptrs[i::self.n]:
                ctx.submit_dev(p, W, H, W)
                nf, nd = ctx.wait()
                f += nf
                d += nd
            self.feats[i], self.descs[i] = f, d
            self.done.wait()

    def step(self):
        self.go.wait()
        self.done.wait()

    def close(self):
        self.stop = True
        self.go.wait()
        for t in self.threads:
            t.join()


ROOT = HERE


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: WORLD_SIZE=%d but --gpus %d; launch with torch.distributed.run" % (world, args.gpus),
                  file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the extraction path has no CPU fallback)")
    # Rehearsal knobs (1-GPU box only): BENCH_FORCE_DEVICE=0 puts every rank on one card and
    # BENCH_BACKEND=gloo swaps RCCL for gloo, to exercise the N>1 control path without N GPUs.
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["BENCH_FORCE_DEVICE"])
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from popsift_amd import _capi as hip
    from popsift_amd.synth import synth

    B = 1 if args.only_roofline else args.batch
    # config 4 seeds 100.. for batches; config 2's own image (seed 2) is image 0 of rank 0
    U = min(B, 16)  # distinct images per rank (0.5 s of host time each to synthesise); the batch cycles through them
    seeds = [2 if (rank == 0 and i == 0) else 100 + rank * U + i for i in range(U)]
    host_imgs = [synth(s, W, H) for s in seeds]
    dev_imgs = [torch.from_numpy(im).cuda(local_rank) for im in host_imgs]  # inputs resident in HBM
    ptrs = [dev_imgs[i % U].data_ptr() for i in range(B)]
    C = max(1, min(args.contexts, B))
    ctxs = [hip.Context(hip.default_params(), device=local_rank) for _ in range(C)]
    workers = Workers(ctxs, ptrs)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if args.only_roofline:
        args.steps, args.warmup = 1, 0
    for _ in range(args.warmup):
        workers.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        workers.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    workers.close()

    feats = sum(workers.feats)   # of one step
    descs = sum(workers.descs)
    rdev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
    cnt = torch.tensor([feats, descs], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    feats_step, descs_step = float(cnt[0].item()), float(cnt[1].item())

    images = args.steps * B * world
    mpix = images * W * H / 1e6
    value = mpix / elapsed

    extra = {}
    if rank == 0 and args.quick:
        print(json.dumps({"value": round(value, 2), "unit": "Mpix/s", "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "steps": args.steps, "quick": True}), flush=True)
    elif rank == 0:
        # ---- single-image device latency (hipEvents, first to last kernel) -------------
        c0 = ctxs[0]
        lat = []
        for _ in range(5):
            c0.submit_dev(ptrs[0], W, H, W)
            c0.wait()
            lat.append(c0.report().ms_device)
        ms_dev = float(np.median(lat))
        rep = c0.report()
        # planes: 6 Gaussian written + read by the next level; DoG on the fly: detection reads the 6 Gaussian planes again
        # (POPSIFT_HIP_DOG_FLY=0: 5 DoG planes written and read instead)
        plane_passes = (2 * 6 + 6) if os.environ.get("POPSIFT_HIP_DOG_FLY", "1") != "0" else (2 * 6 + 2 * 5)
        b_alg = W * H * 1 + 4.0 * rep.pyramid_pixels * plane_passes + 52.0 * rep.ext_total + 512.0 * rep.ori_total
        extra["single_image"] = {
            "ms_device": round(ms_dev, 4), "features": rep.ext_total, "descriptors": rep.ori_total,
            "pipeline_alg_GBps": round(b_alg / (ms_dev * 1e-3) / 1e9, 1),
            "pipeline_frac_of_8TBps": round(b_alg / (ms_dev * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        }
        # ---- roofline of the dominant (HBM-bound) kernel --------------------------------
        # k_blur_tile<HALO,0,64>: the fused "Gaussian level + DoG" launches of the large octaves
        # (5 per such octave; 12 algorithmic bytes per pixel: read plane l-1, write plane l + DoG l-1).
        # Timed with HIP events on the context's own stream (C-ABI profile mode).
        c0.set_profile(True)
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
        c0.set_profile(False)
        if big_n == 0:
            big_ms, big_bytes, big_n = all_ms, all_bytes, all_n
        achieved = big_bytes / (big_ms * 1e-3) / 1e9 if big_ms > 0 else 0.0
        roofline = {
            "kernel": "k_blur_tile<HALO,0,64> (fused H+V Gaussian level, large octaves)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": blur_traffic(),
            "launches": big_n, "avg_launch_us": round(big_ms * 1e3 / max(big_n, 1), 2),
            "alg_bytes_per_launch": round(big_bytes / max(big_n, 1), 1),
            "all_blur_launches": {"launches": all_n, "avg_launch_us": round(all_ms * 1e3 / max(all_n, 1), 2),
                                  "achieved": round(all_bytes / (all_ms * 1e-3) / 1e9, 1) if all_ms > 0 else 0.0},
        }
        # ---- PCIe-inclusive end-to-end rate (host image in, host features out) -----------
        t1 = time.perf_counter()
        n_e2e = 0 if args.only_roofline else 8
        for k in range(n_e2e):
            c0.submit(host_imgs[0])
            c0.fetch()
        if n_e2e:
            extra["host_to_host_single_ctx_mpix_s"] = round(n_e2e * W * H / 1e6 / (time.perf_counter() - t1), 1)

        # the same through the drop-in C++ API (PopSift::enqueue ... SiftJob::get, 4 contexts, pinned result pool):
        # a child process, so that its GPU contexts do not share this one's; reported, never `value`
        exe = os.path.join(ROOT, "popsift_amd", "popsift-bench")
        if world == 1 and not args.only_roofline and os.path.exists(exe):
            import subprocess
            try:
                r = subprocess.run([exe, "--images", "64", "--inflight", "16"], capture_output=True, text=True, timeout=120,
                                   env=dict(os.environ, POPSIFT_CONTEXTS_PER_DEVICE="4", POPSIFT_DEVICES=str(local_rank)))
                extra["host_to_host_cpp_api"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:  # a reported extra: never fail the bench line over it
                extra["host_to_host_cpp_api"] = {"error": str(e)[:200]}

        cpu = None
        if world == 1 and not args.no_cpu_baseline and not args.only_roofline:
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
        out = {
            "metric": "Mpix/s SIFT extract on 1920x1080 (keypoints+descriptors)",
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "1920x1080 u8 grayscale, default popsift::Config (2x upscale, 9 octaves, "
                                   "3 levels, PopSift mode, loop descriptor, RootSift)",
                       "images_per_step_per_gpu": B, "distinct_images_per_gpu": U, "in_flight_contexts_per_gpu": C,
                       "results": "device resident (features + descriptors)"},
            "features_per_s": round(feats_step * args.steps / elapsed, 1),
            "descriptors_per_s": round(descs_step * args.steps / elapsed, 1),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        out.update(extra)
        print(json.dumps(out), flush=True)

    for c in ctxs:
        c.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
