#!/usr/bin/env python3
"""bench.py - env-steps/s of the MI355X-native PPO-over-ALE hot path on synthetic 84x84x4 uint8 batches.

One "step" = one pass of the hot path over one rollout batch: for each of T slots  act (Nature-CNN
forward + categorical sample)  ->  ingest (raw frame pair -> LUT -> 84x84 area resize -> 2-frame max ->
4-frame stack -> rollout slot)  ->  record, then bootstrap + reward-clamp/GAE/returns, then the PPO
update (epochs x minibatches of forward / loss / backward / [RCCL all-reduce] / clip / Adam).
Synthetic inputs are resident in HBM before the timed region.  value = env-steps of all ranks / time,
counting (like the reference, rollout.cc:225) only non-episode-start slots.

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run
(one rank per GPU, RCCL).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

# algorithmic FLOPs per sample of each conv/linear kernel (2*MAC), SURVEY 8(d): fwd 18.69, fwd+bwd 49.52 MFLOP
KFLOP = dict(conv_bwd=2 * (2 * 81 * 64 * 512 + 400 * 32 * 256),  # conv2 dgrad + conv2 wgrad + conv1 wgrad, one launch (bf16 default)
             conv_fwd=2 * (400 * 32 * 256 + 81 * 64 * 512 + 49 * 64 * 576),  # the fused launch (bf16 default) ...
             conv1_fwd=2 * 400 * 32 * 256, conv2_fwd=2 * 81 * 64 * 512, conv3_fwd=2 * 49 * 64 * 576,  # ... or these three
             fc_fwd=2 * 3136 * 512, fc_dgrad=2 * 3136 * 512, fc_wgrad=2 * 3136 * 512, conv3_dgrad=2 * 49 * 64 * 576,
             conv3_wgrad=2 * 49 * 64 * 576, conv2_dgrad=2 * 81 * 64 * 512, conv2_wgrad=2 * 81 * 64 * 512,
             conv1_wgrad=2 * 400 * 32 * 256)
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def kernel_bytes(dtype):
    """algorithmic (minimum) HBM bytes per sample of each kernel: operands read once + result written once
    (DESIGN.md section 4).  e = activation element size; obs is the packed uint8 stack; h is fp32."""
    e = 2 if dtype == "bf16" else 4
    obs, a1, a2, a3, h = 28224, 12800 * e, 5184 * e, 3136 * e, 512 * 4
    dh = 512 * e
    return dict(conv_bwd=a2 + a1 + obs,  # dz2, a1 and the stack read once; dz1 never leaves the CU
                conv_fwd=obs + a1 + a2 + a3,  # a1 / a2 / a3 written once for the backward pass, nothing read back
                conv1_fwd=obs + a1, conv2_fwd=a1 + a2, conv3_fwd=a2 + a3, fc_fwd=a3 + h, fc_dgrad=dh + 2 * a3,
                fc_wgrad=dh + a3, conv3_dgrad=a3 + 2 * a2, conv3_wgrad=a3 + a2, conv2_dgrad=a2 + 2 * a1,
                conv2_wgrad=a2 + a1, conv1_wgrad=a1 + obs)


def log(msg):
    sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] {msg}\n")
    sys.stderr.flush()


def kernel_source_stamp():
    """sha256 (16 hex digits) over the kernel sources: a recorded PMC traffic figure is only quoted for the kernels it
    was measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ale-libtorch-ppo_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


class MappedHost:
    """page-locked host memory the GPU can address (hipHostMallocMapped): what a ring filled by emulator threads looks
    like to the device.  No torch / GPU state is needed to allocate it."""

    def __init__(self, nbytes):
        import ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.ptr = ctypes.c_void_p()
        rc = self.hip.hipHostMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(nbytes), ctypes.c_uint(2))
        if rc != 0:
            raise RuntimeError(f"hipHostMalloc({nbytes}) failed: {rc}")
        self.nbytes = nbytes

    def fill_from(self, arr):
        import ctypes
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes == self.nbytes
        ctypes.memmove(self.ptr, arr.ctypes.data, arr.nbytes)

    @property
    def addr(self):
        return self.ptr.value

    def free(self):
        self.hip.hipHostFree(self.ptr)


def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` outside a launcher: start the one-process-per-GPU job ourselves, as a CHILD and
            # before this process has touched a GPU, and leave with its exit code
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + \
                  sys.argv[1:]
            log("spawning: " + " ".join(cmd))
            raise SystemExit(subprocess.call(cmd))
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch exactly one rank per GPU")
    if "WORLD_SIZE" in os.environ:  # one line per rank: what the launcher handed over
        log(f"rank {rank} local_rank {local_rank} world_size {world} master "
            f"{os.environ.get('MASTER_ADDR', '?')}:{os.environ.get('MASTER_PORT', '?')}")
    if world > 1:
        # a rank holds the null stream, the main, weight-gradient and communication streams plus whatever RCCL and
        # torch.distributed create; the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues,
        # and a communication stream that lands in the main stream's queue would serialise the all-reduce behind the
        # backward pass it is meant to overlap.  Read at runtime initialisation: set before torch loads.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch  # (first: libaleppo.so must bind to the HIP runtime torch ships, not load a second one)
    import torch.distributed as dist
    load_package().device_check(local_rank)  # fails loudly (ALEPPO_ERR_NO_DEVICE) before anything else touches a device
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    pkg = load_package()

    E, T, A, H = args.envs, args.horizon, args.actions, 512
    epochs, M = args.epochs, args.minibatches
    prec = pkg.BF16 if args.dtype == "bf16" else pkg.FP32
    rollp = pkg.ROLLOUT_FP16 if args.rollout_fp16 else pkg.ROLLOUT_FP32
    eng = pkg.Engine(E, T, A, H, precision=prec, device=local_rank, world_size=world, rank=rank, seed=42 + rank,
                     max_minibatch=E * T // M, rollout_precision=rollp)
    if world > 1:
        uid = [pkg.Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(uid[0])
    elif args.force_comm:  # measurement: the data-parallel schedule (early bucket-0 reduce, RCCL calls, no fused tail
        eng.comm_init(pkg.Engine.comm_unique_id())  # reduce) with a 1-rank communicator - its cost without any peer
        eng.set_option(pkg.OPT_FORCE_COMM, 1)
    # random-init weights of the reference architecture (orthogonal init is not reproduced; He-uniform fill)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashfill as hf
    eng.load_params(hf.fill_params(310, H, A))
    if args.update_graph:
        eng.set_option(pkg.OPT_UPDATE_GRAPH, 1)
    lut = (np.arange(256) // 2 * 2).astype(np.uint8)
    eng.set_gray_lut(lut)

    # synthetic raw frame pairs for every slot, resident in HBM: [T][E][2][210][160] uint8 palette codes
    g = torch.Generator(device="cuda")
    g.manual_seed(42 + rank)
    frames = (torch.randint(0, 128, (T, E, 2, 210, 160), device="cuda", generator=g, dtype=torch.int16) * 2).to(
        torch.uint8)
    slot_bytes = E * 2 * 210 * 160
    base_ptr = frames.data_ptr()
    rng = np.random.default_rng(42 + rank)

    # the synthetic emulator outputs of every slot of every rollout are generated BEFORE the timed region
    # (the emulator is out of scope: it stays on host threads); the slot protocol of SURVEY app. A is kept:
    # terminated p=1/200, truncated p=1/2000, each followed by one masked episode-start slot, stale rewards
    def synth_rollout(start0, rewards0):
        rew = np.zeros((T, E), np.float32); te = np.zeros((T, E), np.uint8)
        tr = np.zeros((T, E), np.uint8); st = np.zeros((T, E), np.uint8)
        start, rewards = start0, rewards0
        for t in range(T):
            u = rng.random(E)
            term = ((u < 1 / 200) & (start == 0)).astype(np.uint8)
            trunc = ((u >= 1 / 200) & (u < 1 / 200 + 1 / 2000) & (start == 0)).astype(np.uint8)
            r = np.where(rng.random(E) < 0.05, rng.choice([1.0, 4.0, 7.0], E), 0.0).astype(np.float32)
            rewards = np.where(start == 1, rewards, r).astype(np.float32)
            rew[t], te[t], tr[t], st[t] = rewards, term, trunc, start
            start = (term | trunc).astype(np.uint8)
        return (rew, te, tr, st), start, rewards

    start, rewards = np.ones(E, np.uint8), np.zeros(E, np.float32)
    plans = []
    for _ in range(args.warmup + args.steps + 2):
        plan, start, rewards = synth_rollout(start, rewards)
        plans.append(plan)
    real_steps = 0
    plan_i = 0
    split = [0.0, 0.0, 0.0]  # host wall seconds inside the timed region: act+step loop, finish_rollout, train

    def one_rollout_and_update(count):
        nonlocal real_steps, plan_i
        rew, te, tr, st = plans[plan_i]
        plan_i += 1
        t0 = time.perf_counter()
        if args.python_slot_loop:  # the same T-slot loop driven from Python (ctypes call overhead on every slot)
            ra, ta, ua, sa = rew.ctypes.data, te.ctypes.data, tr.ctypes.data, st.ctypes.data
            for t in range(T):
                eng.act_fast()  # forward + sample; actions land in pinned host memory (an emulator reads them here)
                eng.step_ptr(base_ptr + t * slot_bytes, pkg.DEVICE, pkg.FRAMES_RAW_PAIR, ra + 4 * E * t, ta + E * t,
                             ua + E * t, sa + E * t)
        else:  # native host loop (the reference's Rollout::rollout is C++): act -> actions to pinned host -> step
            eng.replay_rollout(base_ptr, pkg.FRAMES_RAW_PAIR, slot_bytes, rew, te, tr, st)
        t1 = time.perf_counter()
        eng.finish_rollout()
        t2 = time.perf_counter()
        m = eng.train(2.5e-4, epochs, M)
        t3 = time.perf_counter()
        if count:
            real_steps += int((st == 0).sum())
            split[0] += t1 - t0
            split[1] += t2 - t1
            split[2] += t3 - t2
        return m

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"setup done: E={E} T={T} A={A} dtype={args.dtype} world={world}")
    for i in range(args.warmup):
        one_rollout_and_update(False)
        log(f"warmup {i + 1}/{args.warmup}")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = one_rollout_and_update(True)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
    steps_all = torch.tensor([float(real_steps)], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(steps_all, op=dist.ReduceOp.SUM)
    dt = tmax.item()
    value = steps_all.item() / dt
    log(f"timed {args.steps} steps in {dt:.3f} s -> {value:.0f} env-steps/s; per step: slots "
        f"{split[0] / args.steps * 1e3:.2f} ms, finish_rollout {split[1] / args.steps * 1e3:.2f} ms, "
        f"train {split[2] / args.steps * 1e3:.2f} ms")

    # ---- separate profiling passes (HIP events around every launch, on the stream the kernels run on):
    # one rollout (acting shapes) and one update (training shapes) are profiled separately
    roofline = None
    phase = {}
    if True:  # EVERY rank runs the profiled passes (finish_rollout / train hold collectives); rank 0 reports
        def rollout_only():
            rew, te, tr, st = plans[-1]
            ra, ta, ua, sa = rew.ctypes.data, te.ctypes.data, tr.ctypes.data, st.ctypes.data
            for t in range(T):
                eng.act_fast()
                eng.step_ptr(base_ptr + t * slot_bytes, pkg.DEVICE, pkg.FRAMES_RAW_PAIR, ra + 4 * E * t, ta + E * t,
                             ua + E * t, sa + E * t)
            eng.finish_rollout()

        torch.cuda.synchronize()
        t0 = time.perf_counter(); rollout_only(); eng.synchronize(); phase["rollout_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); eng.train(2.5e-4, epochs, M); phase["update_ms"] = (time.perf_counter() - t0) * 1e3
        eng.profile(True)
        eng.profile_reset()
        rollout_only()
        act = {k: eng.profile_read(k) for k in pkg.KERNEL_CLASSES}
        eng.profile_reset()
        eng.train(2.5e-4, epochs, M)  # the timed region's schedule: wgrad kernels + slab reduce on a second stream
        trn = {k: eng.profile_read(k) for k in pkg.KERNEL_CLASSES}
        eng.profile_reset()
        pkg.lib().aleppo_set_option(eng._ctx, pkg.OPT_SERIAL_UPDATE, 1)
        eng.train(2.5e-4, epochs, M)  # every kernel alone on the main stream: its own efficiency
        iso = {k: eng.profile_read(k) for k in pkg.KERNEL_CLASSES}
        pkg.lib().aleppo_set_option(eng._ctx, pkg.OPT_SERIAL_UPDATE, 0)
        eng.profile(False)
        B = E * T // M
        KB = kernel_bytes(args.dtype)
        peak_tf = PEAK_TFLOPS[args.dtype]
        table = {}
        for k in KFLOP:  # both rooflines per launch (B samples); the one with the LARGER minimum time bounds it
            ms = trn[k][0]
            if not ms:
                continue
            tf = KFLOP[k] * B / (ms * 1e-3) / 1e12
            gbs = KB[k] * B / (ms * 1e-3) / 1e9
            t_mfma, t_hbm = KFLOP[k] * B / (peak_tf * 1e12), KB[k] * B / (PEAK_HBM_GBS * 1e9)
            table[k] = dict(ms=round(ms, 4), TFLOPs=round(tf, 1), GBps=round(gbs, 1),
                            bound="hbm" if t_hbm >= t_mfma else "mfma",
                            frac=round(max(t_hbm, t_mfma) / (ms * 1e-3), 4))
        # dominant kernel = the longest one on the update's CRITICAL PATH (main stream: forward chain, dgrad chain, conv1
        # wgrad).  The weight-gradient kernels of the second stream run in the main stream's shadow and stretch with it
        # (fc wgrad: 38 us alone, 72 us beside fc dgrad + conv3 dgrad): their timed-region durations are not a cost.
        main_stream = ("conv_fwd", "conv1_fwd", "conv2_fwd", "conv3_fwd", "fc_fwd", "fc_dgrad", "conv3_dgrad", "conv_bwd", "conv2_dgrad", "conv1_wgrad")
        dom = max((k for k in table if k in main_stream), key=lambda k: trn[k][0] * trn[k][1])
        d = table[dom]
        # HBM traffic of the dominant kernel: recorded by `tests/tools/pmc_traffic.py` (rocprofv3 --pmc passes cannot run
        # inside this process).  The record carries the hash of the kernel sources it was measured on; a record from
        # other sources is NOT quoted (traffic = null).
        pmc, pmc_file = None, None
        stamp = kernel_source_stamp()
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if name.endswith("_pmc_traffic.json") and args.dtype == "bf16" and B == 4096:
                rec = json.load(open(os.path.join(ROOT, "profiles", name)))
                if rec.get("kernel_source_sha16") == stamp and dom in rec:
                    pmc, pmc_file = rec[dom], name
                    break
        if d["bound"] == "hbm":
            roofline = dict(bound="hbm", kernel=dom, achieved=d["GBps"], peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=round(d["GBps"] / PEAK_HBM_GBS, 4),
                            traffic=(pmc["traffic_MB"] * 1e6 if pmc else None),
                            algorithmic_bytes_per_launch=KB[dom] * B)
        else:
            roofline = dict(bound="mfma", kernel=dom, achieved=d["TFLOPs"], peak=peak_tf, unit="TFLOP/s",
                            frac=round(d["TFLOPs"] / peak_tf, 4), traffic=(pmc["traffic_MB"] * 1e6 if pmc else None),
                            flop_per_launch=KFLOP[dom] * B)
        roofline.update(avg_launch_ms=round(trn[dom][0], 4), launches=trn[dom][1],
                        traffic_recorded_in=(f"profiles/{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                             f"passes, FETCH x2 gfx950 correction; kernel sources {stamp})") if pmc
                        else None, kernel_source_sha16=stamp)
        upd_ms = sum(v[0] * v[1] for v in trn.values())
        roofline["update_all_kernels_TFLOPs"] = round(sum(KFLOP[k] for k in table) * B * epochs * M / (upd_ms * 1e-3) / 1e12, 2)
        roofline["update_all_kernels_GBps"] = round(sum(KB[k] for k in table) * B * epochs * M / (upd_ms * 1e-3) / 1e9, 1)
        roofline["update_kernels"] = table
        iso_tab = {}
        for k in KFLOP:  # the same table with every kernel running alone (ALEPPO_OPT_SERIAL_UPDATE)
            ms = iso[k][0]
            if ms:
                t_mfma, t_hbm = KFLOP[k] * B / (peak_tf * 1e12), KB[k] * B / (PEAK_HBM_GBS * 1e9)
                iso_tab[k] = dict(ms=round(ms, 4), GBps=round(KB[k] * B / (ms * 1e-3) / 1e9, 1),
                                  TFLOPs=round(KFLOP[k] * B / (ms * 1e-3) / 1e12, 1),
                                  frac=round(max(t_hbm, t_mfma) / (ms * 1e-3), 4))
        roofline["update_kernels_isolated"] = iso_tab
        roofline["isolated"] = dict(kernel=dom, avg_launch_ms=iso_tab[dom]["ms"], frac=iso_tab[dom]["frac"],
                                    achieved=iso_tab[dom]["GBps"] if d["bound"] == "hbm" else iso_tab[dom]["TFLOPs"],
                                    note="same kernel alone on the GPU; in the timed region it shares the GPU with the "
                                         "kernels co-scheduled on the second stream")
        roofline["update_other_kernel_ms"] = {k: round(v[0], 4) for k, v in trn.items() if v[1] and k not in KFLOP}
        roofline["acting_kernel_ms_per_step"] = {k: round(v[0], 4) for k, v in act.items() if v[1]}
        # frame ingest: its own kernel, or fused in front of the acting convolutions (then the launch also does conv1-3)
        ing_ms = act["ingest"][0] or act["act_fused"][0] or float("nan")
        roofline["hbm_kernels"] = dict(
            ingest_GBps=round(74256 * E / (ing_ms * 1e-3) / 1e9, 1), ingest_frac=round(74256 * E / (ing_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            ingest_kernel="ingest_kernel" if act["ingest"][1] else "act_conv_kernel<2> (ingest + conv1-3 in one launch)",
            gae_GBps=round(23 * E * T / (act["gae"][0] * 1e-3) / 1e9, 2),
            adam_GBps=round(28 * eng.param_count / (trn["adam"][0] * 1e-3) / 1e9, 1),
            head_GBps=round((97 + 2 * 512 * 4) * B / (trn["head"][0] * 1e-3) / 1e9, 1))
        roofline["phase_wall_ms"] = {k: round(v, 3) for k, v in phase.items()}

    # ---- the fused tail of the backward pass (conv2 dgrad + conv2 wgrad + conv1 wgrad in one launch, the default;
    # ALEPPO_OPT_FUSED_BWD = 0: three launches on two streams; csrc/conv_bwd_fused.hpp): the same update both ways,
    # alternating, same process
    fused_bwd = None
    if rank == 0 and world == 1 and args.dtype == "bf16" and hasattr(pkg, "OPT_FUSED_BWD") and not args.no_host_legs:
        t_on, t_off = [], []
        for rep in range(4):
            for on, acc in ((2, t_on), (0, t_off)):
                eng.set_option(pkg.OPT_FUSED_BWD, on)
                eng.train(2.5e-4, epochs, M)
                eng.synchronize()
                t0 = time.perf_counter()
                eng.train(2.5e-4, epochs, M)
                eng.synchronize()
                acc.append((time.perf_counter() - t0) * 1e3)
        eng.set_option(pkg.OPT_FUSED_BWD, 1)  # (the default)
        fused_bwd = dict(update_ms=round(sorted(t_on)[len(t_on) // 2], 3), update_ms_three_launches=round(sorted(t_off)[len(t_off) // 2], 3),
                         note="update wall time with the fused backward tail (default) / with the three launches it replaces "
                              "(median of 4, alternating); 263 instead of 619 MB per minibatch (DESIGN.md 4e)")

    # ---- the SAME workload with the frames where a host emulator leaves them (rollout.cc:325-326): page-locked host
    # memory the ingest kernel reads in place over PCIe (ALEPPO_HOST_MAPPED).  Reported beside `value`, never as it.
    host_legs = None
    if rank == 0 and world == 1 and not args.no_host_legs:
        host_legs = {}
        hsteps = max(2, min(args.steps, 5))
        # (+ the metric's own wording, "84x84x4 uint8 batches": finished 84x84 frames resident in HBM, no preprocessing)
        for leg, kind, per_env, loc in (("frames_84_hbm", pkg.FRAMES_84, 84 * 84, pkg.DEVICE),
                                        ("frames_84_mapped_host", pkg.FRAMES_84, 84 * 84, pkg.HOST_MAPPED),
                                        ("raw_pair_mapped_host", pkg.FRAMES_RAW_PAIR, 2 * 210 * 160, pkg.HOST_MAPPED)):
            try:
                f84 = np.random.default_rng(7).integers(0, 256, (T, E, 84, 84), dtype=np.uint8)
                if loc == pkg.DEVICE:
                    class _Dev:  # same interface as MappedHost, frames in HBM
                        def __init__(self, arr):
                            self.t = torch.from_numpy(arr).cuda()
                            self.addr = self.t.data_ptr()

                        def free(self):
                            del self.t
                    mh = _Dev(f84)
                else:
                    mh = MappedHost(T * E * per_env)
                    mh.fill_from(frames.cpu().numpy() if kind == pkg.FRAMES_RAW_PAIR else f84)
                rs = 0
                plan_j = 0
                for i in range(1 + hsteps):
                    rew, te, tr, st = plans[plan_j % len(plans)]
                    plan_j += 1
                    if i == 1:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    eng.replay_rollout(mh.addr, kind, E * per_env, rew, te, tr, st, location=loc)
                    eng.finish_rollout()
                    eng.train(2.5e-4, epochs, M)
                    if i >= 1:
                        rs += int((st == 0).sum())
                torch.cuda.synchronize()
                hdt = time.perf_counter() - t0
                mh.free()
                host_legs[leg] = {"value": round(rs / hdt, 1), "unit": "env-steps/s", "steps": hsteps,
                                  "ms_per_step": round(hdt / hsteps * 1e3, 3),
                                  "pcie_bytes_per_slot": E * per_env if loc != pkg.DEVICE else 0}
                log(f"frame leg {leg}: {rs / hdt:.0f} env-steps/s")
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"host-frames leg {leg} failed: {e}\n")

    # ---- CPU baseline beside it (rank 0, N=1 only): the reference's own compiled CPU-libtorch path
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(E, T, A, epochs, M)

    last_loss, last_norm = float(metrics["loss"][-1, -1]), float(metrics["grad_norm"][-1, -1])
    eng.close()
    # (the main engine is closed first: a second live context would share the runtime's 4 hardware queues with it, and the
    # v1 engine's two streams can then land in ONE queue - 3.14 instead of 3.6 M env-steps/s)
    # ---- secondary leg: the reference's configs/v1.yaml shape (4096 envs x T=5, 1 epoch, 16 minibatches of 1280) -
    # the configuration its published ~26 k env-steps/s was quoted on.  N=1 only, reported beside `value`.
    v1 = None
    if rank == 0 and world == 1 and not args.no_v1:
        try:
            v1 = v1_shape_leg(pkg, args, local_rank)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"v1-shape leg failed: {e}\n")

    if rank == 0:
        out = {
            "metric": "env steps/sec (whole node), Breakout 84x84x4", "value": round(value, 1),
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Breakout-shaped synthetic, {E} envs/GPU x T={T}, A={A}, H=512, "
                                   f"{epochs} epochs x {M} minibatches of {E * T // M} (configs[1], v0.yaml update shape)",
                       "envs_per_gpu": E, "horizon": T, "epochs": epochs, "minibatches": M,
                       "parallelism": f"dp{world}", "frames": "raw u8 [E,2,210,160] pairs resident in HBM",
                       "rollout_planes": "fp16" if args.rollout_fp16 else "fp32",
                       "update_graph": bool(args.update_graph),
                       "dp_schedule_forced": bool(args.force_comm and world == 1)},
            "roofline": roofline, "cpu_baseline": cpu,
            "frames_84_hbm": (host_legs or {}).pop("frames_84_hbm", None), "host_frames": host_legs,
            "vs_reference_published_v1_26289": round(value / 26289.0, 2), "v1_shape": v1,
            "fused_bwd_option": fused_bwd,
            "last_loss": last_loss, "last_grad_norm": last_norm,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def v1_shape_leg(pkg, args, device):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashfill as hf
    E, T, A, H, epochs, M = 4096, 5, 4, 512, 1, 16
    prec = pkg.BF16 if args.dtype == "bf16" else pkg.FP32
    eng = pkg.Engine(E, T, A, H, precision=prec, device=device, clip_param=0.2, value_loss_coef=0.4, seed=7,
                     max_minibatch=E * T // M)
    eng.load_params(hf.fill_params(310, H, A))
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    frames = torch.randint(0, 256, (T, E, 84, 84), device="cuda", generator=g, dtype=torch.int16).to(torch.uint8)
    base, slot = frames.data_ptr(), E * 84 * 84
    rng = np.random.default_rng(7)
    rew = np.where(rng.random((T, E)) < 0.05, 1.0, 0.0).astype(np.float32)
    z = np.zeros((T, E), np.uint8)
    st0 = np.ones((T, E), np.uint8)
    st0[1:] = 0
    steps = 12

    def one(first):
        st = st0 if first else z
        ra, za, sa = rew.ctypes.data, z.ctypes.data, st.ctypes.data
        for t in range(T):
            eng.act_fast()
            eng.step_ptr(base + t * slot, pkg.DEVICE, pkg.FRAMES_84, ra + 4 * E * t, za + E * t, za + E * t, sa + E * t)
        eng.finish_rollout()
        eng.train(2.5e-4, epochs, M)

    one(True)
    for _ in range(3):
        one(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one(False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.close()
    return {"value": round(E * T * steps / dt, 1), "unit": "env-steps/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "workload": "4096 envs x T=5, 1 epoch x 16 minibatches of 1280, pre-resized 84x84 frames in HBM",
            "vs_reference_published_v1_26289": round(E * T * steps / dt / 26289.0, 2)}


def cpu_baseline(E, T, A, epochs, M):
    """time oracle/_ref/ref_harness (the reference's own gae.cc / buffer.cc / losses.cc / train.{h,cc} compiled against
    CPU libtorch) on the GPU box's host cores, on a bounded sample of the SAME workload as `value` (same envs, horizon,
    minibatches: one rollout + update after one warm-up, ~20-40 s), with the reference's own configs/v0.yaml shape
    (8 envs) beside it."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    # the GPU box gives one GPU a share of 16 host cores; never oversubscribe beyond the affinity mask
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    log(f"cpu baseline: reference harness on {threads} threads")

    def harness(e, t, a, ep, m, iters):
        out = subprocess.run([exe, "bench", str(e), str(t), "512", str(a), str(ep), str(m), str(iters), str(threads)],
                             check=True, capture_output=True, text=True, timeout=400).stdout.strip().splitlines()[-1]
        return json.loads(out)

    if os.path.exists(exe):
        try:
            j = harness(E, T, A, epochs, M, 1)
            res = {"value": round(j["env_steps_per_s"], 1), "unit": "env-steps/s", "cores": threads,
                   "kind": "reference",
                   "sample": f"the workload of `value`: {E} envs x T={T}, {epochs} epochs x {M} minibatches of "
                             f"{E * T // M}, H=512, A={A}; 1 rollout+update after 1 warm-up ({j['seconds']:.1f} s), "
                             "libtorch CPU, pre-resized 84x84 frames"}
            try:
                v0 = harness(8, 128, 4, 4, 4, 12)
                res["v0_yaml_shape"] = {"value": round(v0["env_steps_per_s"], 1), "unit": "env-steps/s",
                                        "sample": "configs/v0.yaml: 8 envs x T=128, 4 epochs x 4 minibatches of 256; "
                                                  f"12 rollouts+updates after 1 warm-up ({v0['seconds']:.1f} s)"}
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"v0-shape cpu point failed: {e}\n")
            return res
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"reference cpu baseline failed ({e}); falling back to the C port\n")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashfill as hf
    import oracle_lib as orc
    H, N, Mp, ep = 512, 1024, 1, 1
    params = hf.fill_params(310, H, A)
    obs = hf.hf_bytes(5, (N, 4, 84, 84))
    t0 = time.perf_counter()
    orc.train(params, H, A, obs, np.zeros(N, np.int64), orc.log_softmax(np.zeros((N, A), np.float32)),
              np.ones(N, np.float32), np.ones(N, np.float32), np.ones(N, np.uint8), ep, Mp)
    dt = time.perf_counter() - t0
    return {"value": round(N / dt / epochs, 1), "unit": "env-steps/s", "cores": orc.lib().oracle_num_threads(),
            "kind": "port", "sample": f"oracle C port: {N}-sample x 1 epoch update, scaled to {epochs} epochs "
                                      "(update only; acting forward not included)"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=128, help="environments per GPU (configs[1]: 128)")
    ap.add_argument("--horizon", type=int, default=128)
    ap.add_argument("--actions", type=int, default=4)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--minibatches", type=int, default=4)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--python-slot-loop", action="store_true",
                    help="drive the per-slot act/step loop from Python instead of aleppo_replay_rollout")
    ap.add_argument("--no-v1", action="store_true", help="skip the secondary v1.yaml-shape leg")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the frames-in-pinned-host-memory legs")
    ap.add_argument("--rollout-fp16", action="store_true",
                    help="BASELINE configs[4]: half-precision rollout planes (use with --envs 256 --actions 6)")
    ap.add_argument("--update-graph", type=int, default=0,
                    help="1: replay the update loop as a captured hipGraph (the reference's `cuda_graph: true`)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N=1 only: run aleppo_train's data-parallel schedule through a 1-rank RCCL communicator")
    ap.add_argument("--master-port", type=int, default=29517, help="rendezvous port when bench.py spawns the ranks")
    run(ap.parse_args())
